#!/bin/bash
# Build an experiment variant of the library: tools/build_exp.sh NAME "-DMGF_EXP=3 ..." [source.hip]  -> exp_build/libmgf_NAME.so
# (morphganformer_amd/build.py: every object is compiled into exp_build/_obj under a name that hashes its flags, compiler and source;
# nothing is copied from -- or can leak into -- the product's object cache)
set -e
cd "$(dirname "$0")/.."
exec python -m morphganformer_amd.build --exp "$1" --flags "-DMGF_TUNING_HOOKS $2" --source "${3:-conv_taps.hip}"
