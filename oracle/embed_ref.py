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


# --------------------------------------------------------------------------------------------------------------------------
# FaceNet: facenet_pytorch.InceptionResnetV1 -- the embedder 1024_example_FaceNet_percept.py:30-32,147-158 calls.
# PARITY UNPINNED: the package (un-vendored, not in requirements.txt) and its weights are absent offline, and the reference holds
# no fixture of it; this restates the package's PUBLISHED module definitions (facenet_pytorch/models/inception_resnet_v1.py:
# BasicConv2d, Block35, Block17, Block8, Mixed_6a, Mixed_7a, InceptionResnetV1.forward with classify=False) with torch's own ops,
# on a state_dict under the package's key names.

def _basic(sd, name, x, stride=1, padding=0):
    """BasicConv2d.forward: Conv2d(bias=False) -> BatchNorm2d(eps=0.001) -> ReLU."""
    x = F.conv2d(x, sd[name + ".conv.weight"], None, stride=stride, padding=padding)
    x = F.batch_norm(x, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"], sd[name + ".bn.weight"], sd[name + ".bn.bias"],
                     training=False, eps=0.001)
    return F.relu(x)


def _block35(sd, p, x, scale):
    x0 = _basic(sd, p + ".branch0", x)
    x1 = _basic(sd, p + ".branch1.1", _basic(sd, p + ".branch1.0", x), padding=1)
    x2 = _basic(sd, p + ".branch2.2", _basic(sd, p + ".branch2.1", _basic(sd, p + ".branch2.0", x), padding=1), padding=1)
    out = F.conv2d(torch.cat((x0, x1, x2), 1), sd[p + ".conv2d.weight"], sd[p + ".conv2d.bias"])
    return F.relu(out * scale + x)


def _block17(sd, p, x, scale):
    x0 = _basic(sd, p + ".branch0", x)
    x1 = _basic(sd, p + ".branch1.0", x)
    x1 = _basic(sd, p + ".branch1.1", x1, padding=(0, 3))
    x1 = _basic(sd, p + ".branch1.2", x1, padding=(3, 0))
    out = F.conv2d(torch.cat((x0, x1), 1), sd[p + ".conv2d.weight"], sd[p + ".conv2d.bias"])
    return F.relu(out * scale + x)


def _block8(sd, p, x, scale, no_relu=False):
    x0 = _basic(sd, p + ".branch0", x)
    x1 = _basic(sd, p + ".branch1.0", x)
    x1 = _basic(sd, p + ".branch1.1", x1, padding=(0, 1))
    x1 = _basic(sd, p + ".branch1.2", x1, padding=(1, 0))
    out = F.conv2d(torch.cat((x0, x1), 1), sd[p + ".conv2d.weight"], sd[p + ".conv2d.bias"])
    out = out * scale + x
    return out if no_relu else F.relu(out)


def inception_resnet_v1_ref(sd, x, taps=None):
    """InceptionResnetV1.forward, eval mode, classify=False: x [n,3,H,W] -> unit-norm embeddings [n,512]."""
    x = _basic(sd, "conv2d_1a", x, stride=2)
    x = _basic(sd, "conv2d_2a", x)
    x = _basic(sd, "conv2d_2b", x, padding=1)
    x = F.max_pool2d(x, 3, stride=2)
    x = _basic(sd, "conv2d_3b", x)
    x = _basic(sd, "conv2d_4a", x)
    x = _basic(sd, "conv2d_4b", x, stride=2)
    if taps is not None:
        taps["stem"] = x
    for r in range(5):
        x = _block35(sd, f"repeat_1.{r}", x, 0.17)
    if taps is not None:
        taps["repeat_1"] = x
    p = "mixed_6a"
    x = torch.cat((_basic(sd, p + ".branch0", x, stride=2),
                   _basic(sd, p + ".branch1.2", _basic(sd, p + ".branch1.1", _basic(sd, p + ".branch1.0", x), padding=1), stride=2),
                   F.max_pool2d(x, 3, stride=2)), 1)
    for r in range(10):
        x = _block17(sd, f"repeat_2.{r}", x, 0.10)
    if taps is not None:
        taps["repeat_2"] = x
    p = "mixed_7a"
    x = torch.cat((_basic(sd, p + ".branch0.1", _basic(sd, p + ".branch0.0", x), stride=2),
                   _basic(sd, p + ".branch1.1", _basic(sd, p + ".branch1.0", x), stride=2),
                   _basic(sd, p + ".branch2.2", _basic(sd, p + ".branch2.1", _basic(sd, p + ".branch2.0", x), padding=1), stride=2),
                   F.max_pool2d(x, 3, stride=2)), 1)
    for r in range(5):
        x = _block8(sd, f"repeat_3.{r}", x, 0.20)
    x = _block8(sd, "block8", x, 1.0, no_relu=True)
    if taps is not None:
        taps["block8"] = x
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)                       # avgpool_1a; dropout = identity in eval
    x = F.linear(x, sd["last_linear.weight"], None)
    x = F.batch_norm(x, sd["last_bn.running_mean"], sd["last_bn.running_var"], sd["last_bn.weight"], sd["last_bn.bias"], training=False, eps=0.001)
    return F.normalize(x, p=2, dim=1)


def facenet_loss_ref(sd, pred, target):
    """MSE(model(img_gen), model(target)) on the flattened embeddings, the images un-resized (1024_example_FaceNet_percept.py:147-158)."""
    return ((inception_resnet_v1_ref(sd, pred) - inception_resnet_v1_ref(sd, target)) ** 2).mean(dim=1)
