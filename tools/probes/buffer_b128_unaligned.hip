// Probe (round 6): 16-byte raw buffer loads / stores at 4-byte alignment, and a 16-byte access that straddles the end of the resource --
// which of its dwords come back, which are written?  hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_b128_unaligned.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k(const float* g, float* out, float* st, int records, int shift) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, records * 4, 0x00020000);
    const unsigned off = (threadIdx.x * 4u + (unsigned)shift) * 4u;          // element offset 4 lane + shift: 4-byte aligned only when shift % 4 != 0
    const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, records * 4, 0x00020000);
    v4u o;
    o.x = __builtin_bit_cast(unsigned, 1000.f + threadIdx.x * 4 + 0); o.y = __builtin_bit_cast(unsigned, 1000.f + threadIdx.x * 4 + 1);
    o.z = __builtin_bit_cast(unsigned, 1000.f + threadIdx.x * 4 + 2); o.w = __builtin_bit_cast(unsigned, 1000.f + threadIdx.x * 4 + 3);
    __builtin_amdgcn_raw_buffer_store_b128(o, w, off, 0, 0);
}
int main() {
    float *g, *o, *s; hipMalloc(&g, 4096 * 4); hipMalloc(&o, 4096 * 4); hipMalloc(&s, 4096 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i + 0.5f;
    for (int shift = 0; shift < 4; ++shift) for (int records : {253, 254, 255, 256}) {
        hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(s, 0, 4096 * 4);
        k<<<1, 64>>>(g, o, s, records, shift);
        float r[256], t[4096]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost); hipMemcpy(t, s, sizeof(t), hipMemcpyDeviceToHost);
        int bad_inner = 0, tail_read = 0, tail_zero = 0, bad_store = 0, tail_stored = 0, past = 0;
        for (int i = 0; i < 256; ++i) {
            const int e = i + shift;                                          // element this slot addresses
            const int lane_first = (i / 4) * 4 + shift;
            const bool straddles = lane_first < records && lane_first + 3 >= records;
            if (e < records && !straddles) { if (r[i] != e + 0.5f) ++bad_inner; if (t[e] != 1000.f + i) ++bad_store; }
            else if (e < records) { if (r[i] == e + 0.5f) ++tail_read; else if (r[i] == 0.f) ++tail_zero; if (t[e] == 1000.f + i) ++tail_stored; }
        }
        for (int e = records; e < records + 8; ++e) if (t[e] != 0.f) ++past;
        printf("shift %d records %d: inner loads bad %d, inner stores bad %d | in-range dwords of the straddling vector: read %d zero %d stored %d | written past the end %d\n",
               shift, records, bad_inner, bad_store, tail_read, tail_zero, tail_stored, past);
    }
    return 0;
}
