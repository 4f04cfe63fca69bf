"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the reference's IResNet embedder
(backbones/iresnet.py) and of the embedding-MSE biometric loss (1024_example_FaceNet_percept.py:147-158).

Plain torch.nn.functional on a state_dict with the reference's key names, eval mode (BatchNorm on running statistics,
dropout = identity).  Pinned against the reference module itself: tests/golden/iresnet18.npz is produced by loading the same
seeded state into `backbones.iresnet.iresnet18()` (oracle/make_golden.py::gold_iresnet).
"""
import torch
import torch.nn.functional as F

from morphganformer_amd.iresnet import block_table      # the table of (prefix, inplanes, planes, stride, downsample) rows only


def _bn(sd, name, x):
    """nn.BatchNorm2d / BatchNorm1d(eps=1e-5).eval()  (iresnet.py:37,41,44,76,96,100)"""
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        training=False, eps=1e-5)


def iresnet_ref(sd, x, depth=50, taps=None):
    """IResNet.forward (iresnet.py:145-160) with IBasicBlock.forward (:46-58).  x: [n,3,112,112] float32."""
    x = F.conv2d(x, sd["conv1.weight"], None, stride=1, padding=1)
    x = F.prelu(_bn(sd, "bn1", x), sd["prelu.weight"])
    for p, inpl, planes, stride, ds in block_table(depth):
        identity = x
        out = _bn(sd, p + ".bn1", x)
        out = F.conv2d(out, sd[p + ".conv1.weight"], None, stride=1, padding=1)
        out = F.prelu(_bn(sd, p + ".bn2", out), sd[p + ".prelu.weight"])
        out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride=stride, padding=1)
        out = _bn(sd, p + ".bn3", out)
        if ds:
            identity = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride))
        x = out + identity
        if taps is not None:
            taps[p] = x
    x = _bn(sd, "bn2", x)
    x = torch.flatten(x, 1)
    x = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    return _bn(sd, "features", x)


def resize112_ref(img):
    return F.interpolate(img, size=(112, 112), mode="bilinear", align_corners=False)


def biometric_loss_ref(sd, pred, target, depth=50):
    """MSE(model(img_gen), model(target)) on the flattened embeddings (1024_example_FaceNet_percept.py:147-158)."""
    e0 = iresnet_ref(sd, resize112_ref(pred), depth)
    e1 = iresnet_ref(sd, resize112_ref(target), depth)
    return ((e0 - e1) ** 2).mean(dim=1)
