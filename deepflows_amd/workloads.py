"""The reference's hot-path workloads, restated on this package's nn API: the model classes
its training scripts define and the per-batch step they run.

  LinearRegressionModel  test/LinearRegression.py:27-33
  MLP                    BASELINE config 784->256->10 (script variant 784->100->20->10:
                         test/MLP_MNIST_cuda.py:44-56)
  CIFAR10_CNN            test/CNN_CIFAR10_cuda.py:61-108
  ResidualBlock / ResNet18   test/ResNet_CIFAR10_cuda.py:20-100 (channels 32/64/128/256,
                         max-pool after the stem, NO ReLU inside the blocks, global average
                         pool written as two `mean(axis=2)` calls)
  ResNet18Wide           test/ResNet.py:24-154 (the repository's OTHER ResNet-18: channels 64/128/256/512, no
                         max-pool after the stem, ReLU AFTER the residual add, still none inside the block)
  VGG16                  test/VGG.py:7-152
  ConvBlock / MobileNetV1    test/MobileNet.py:18-108
  train_step             the loop body of test/ResNet_CIFAR10_cuda.py:185-190

`register_blocks`: the reference keeps its residual blocks in plain Python lists, which
nn.Module does not register, so `model.parameters()` yields 5 of the 62 tensors and the
optimizer never touches the blocks.  register_blocks=False reproduces that
exactly; register_blocks=True (default) registers every block so all 2 797 610 parameters are
trained and all-reduced.
"""
import numpy as np

from . import nn, tensor
from .tensor import Graph, Tensor, realize


class LinearRegressionModel(nn.Module):
    def __init__(self, device="hip"):
        super().__init__()
        self.linear = nn.Linear(1, 1, device=device)

    def forward(self, x):
        return self.linear(x)


class MLP(nn.Module):
    def __init__(self, sizes=(784, 256, 10), device="hip"):
        super().__init__()
        self.n_layers = len(sizes) - 1
        for i in range(self.n_layers):
            setattr(self, f"fc{i + 1}", nn.Linear(sizes[i], sizes[i + 1], device=device))

    def forward(self, x):
        # test/MLP_MNIST_cuda.py:51-54: `x = F.relu(self.fc1(x)) ... x = self.fc3(x)` -- on the hip device relu(fc(x)) becomes one
        # node inside the modules (nn/lazy.py)
        for i in range(self.n_layers):
            x = getattr(self, f"fc{i + 1}")(x)
            if i + 1 < self.n_layers:
                x = nn.functional.relu(x)
        return x


class CIFAR10_CNN(nn.Module):
    def __init__(self, num_classes=10, device="hip"):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 32, kernel_size=5, padding=2, device=device)
        self.bn1 = nn.BatchNorm2d(32, device=device)
        self.relu1 = nn.ReLU()
        self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=5, padding=2, device=device)
        self.bn2 = nn.BatchNorm2d(64, device=device)
        self.relu2 = nn.ReLU()
        self.pool2 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv3 = nn.Conv2d(64, 128, kernel_size=3, padding=1, device=device)
        self.bn3 = nn.BatchNorm2d(128, device=device)
        self.relu3 = nn.ReLU()
        self.pool3 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.drop = nn.Dropout(0.5)
        self.fc = nn.Linear(128 * 4 * 4, num_classes, device=device)

    def forward(self, x):
        # the script's text (test/CNN_CIFAR10_cuda.py:86-108); on the hip device the modules record their calls and launch
        # pool(relu(bn(conv(x)))) as the stem node / the fused BatchNorm-ReLU-pool node (nn/lazy.py)
        x = self.conv1(x)
        x = self.bn1(x)
        x = self.relu1(x)
        x = self.pool1(x)

        x = self.conv2(x)
        x = self.bn2(x)
        x = self.relu2(x)
        x = self.pool2(x)

        x = self.conv3(x)
        x = self.bn3(x)
        x = self.relu3(x)
        x = self.pool3(x)

        x = x.reshape(x.shape[0], -1)
        x = self.drop(x)
        x = self.fc(x)
        return x


class ResidualBlock(nn.Module):
    """`relu_after_add`: test/ResNet.py:39-62 ends the block with `relu(out + identity)`; the CIFAR script's block
    (test/ResNet_CIFAR10_cuda.py:31-47) has no activation at all."""

    def __init__(self, in_channels, out_channels, stride=1, downsample=None, device="hip", relu_after_add=False):
        super().__init__()
        self.relu_after_add = relu_after_add
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1,
                               bias=False, device=device)
        self.bn1 = nn.BatchNorm2d(out_channels, device=device)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1,
                               bias=False, device=device)
        self.bn2 = nn.BatchNorm2d(out_channels, device=device)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        ds = self.downsample
        pair = ds is not None and len(ds) == 2 and isinstance(ds[1], nn.BatchNorm2d)
        z1 = None
        if pair:
            # conv1 and the shortcut convolution read the same tensor: one launch for both where the device has it
            both = None
            if bool(getattr(x.device.mod, "fused", False)) and isinstance(ds[0], nn.Conv2d):
                both = nn.functional._Conv2dFused.pair_s2(x, self.conv1, ds[0])
            if both is not None:
                z1, zd = both
            else:
                zd = realize(ds[0](x))         # the shortcut convolution; its BatchNorm joins bn2 below
        elif ds is not None:
            for layer in ds:
                identity = layer(identity)
        if z1 is None:
            z1 = realize(self.conv1(x))      # (this forward() names the fused entry points itself: no recorded calls)
        if pair and hasattr(z1, "shortcut_sibling") and hasattr(zd, "shortcut_sibling"):
            z1.shortcut_sibling = zd           # both read x: one launch forms both input gradients (nn.functional)
        # conv2(bn1(z1)): bn1 applied on conv2's operand loads (one node, bn1's output is never written); None when
        # that form does not apply here
        z2 = nn.functional.bn_conv2d_train(z1, self.bn1, self.conv2)
        if z2 is None:
            z2 = realize(self.conv2(self.bn1(z1)))
        if pair:
            if nn.functional.batch_norm2d_pair_fusable(z2, self.bn2, zd, ds[1]):
                # bn2(conv2) + bn_d(conv_d) as one node: one launch each way for the two BatchNorms
                out = nn.functional.batch_norm2d_pair_train(z2, self.bn2, zd, ds[1])
                return nn.functional.relu(out) if self.relu_after_add else out
            identity = ds[1](zd)
        # bn2(...) + identity; one kernel on a fused device, the reference's "out + identity" otherwise
        out = self.bn2(z2, residual=identity)
        return nn.functional.relu(out) if self.relu_after_add else out


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=10, device="hip", register_blocks=True,
                 widths=(32, 64, 128, 256), stem_pool=True, relu_after_add=False):
        super().__init__()
        self._dev = device
        self._register = register_blocks
        self._stem_pool, self._relu_after_add = stem_pool, relu_after_add
        self.in_channels = widths[0]
        self.conv1 = nn.Conv2d(3, widths[0], kernel_size=3, stride=1, padding=1, bias=False, device=device)
        self.bn1 = nn.BatchNorm2d(widths[0], device=device)
        self.relu = nn.ReLU()
        self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.layer1 = self._make_layer(block, widths[0], layers[0], stride=1)
        self.layer2 = self._make_layer(block, widths[1], layers[1], stride=2)
        self.layer3 = self._make_layer(block, widths[2], layers[2], stride=2)
        self.layer4 = self._make_layer(block, widths[3], layers[3], stride=2)
        self.fc = nn.Linear(widths[3], num_classes, device=device)

    def _make_layer(self, block, out_channels, blocks, stride=1):
        downsample = None
        if stride != 1 or self.in_channels != out_channels:
            conv = nn.Conv2d(self.in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                             device=self._dev)
            bn = nn.BatchNorm2d(out_channels, device=self._dev)
            downsample = nn.ModuleList([conv, bn]) if self._register else [conv, bn]
        kw = {"relu_after_add": True} if self._relu_after_add else {}
        layers = [block(self.in_channels, out_channels, stride, downsample, device=self._dev, **kw)]
        self.in_channels = out_channels
        for _ in range(1, blocks):
            layers.append(block(self.in_channels, out_channels, device=self._dev, **kw))
        return nn.ModuleList(layers) if self._register else layers

    def forward(self, x):
        if not isinstance(x, Tensor):
            x = Tensor(x)
        stem = nn.functional.stem_conv_bn_relu_pool_train(x, self.conv1, self.bn1, self.pool1) if self._stem_pool else None
        if stem is not None:
            x = stem                                                  # one node, the 33 MB convolution output never exists
        elif not self._stem_pool:
            x = self.relu(self.bn1(self.conv1(x)))                    # test/ResNet.py:110-112: no pooling
        else:
            x = realize(self.conv1(x))
            if nn.functional.bn_relu_maxpool2_fusable(x, self.bn1, self.pool1):
                x = nn.functional.bn_relu_maxpool2_train(x, self.bn1)     # one node on the hip device
            else:
                x = self.pool1(self.relu(self.bn1(x)))
        for stage in (self.layer1, self.layer2, self.layer3):
            for blk in stage:
                x = blk(x)
        # everything created after this tensor is layer4 + the classifier: 2.4 M of the 2.8 M
        # parameters, whose gradients are final when the backward sweep gets back here
        self.backward_milestone = x
        for blk in self.layer4:
            x = blk(x)
        # mean(axis=2) twice (each divides by the TOTAL element count), then the classifier: one node on
        # the hip device -- recorded, so that a CrossEntropyLoss can complete it (head + loss from one launch, nn/lazy.py)
        return nn.lazy.head(x, self.fc)


# ---- the other convolutional models of the example scripts -----------------------
_VGG16_STAGES = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))


class VGG16(nn.Module):
    """test/VGG.py:7-152: 13 x (3x3 conv, no bias -> BatchNorm2d -> ReLU), a 2x2 max-pool after each of
    the five stages, then Linear(512*s*s, 4096) - ReLU - Dropout - Linear(4096, 4096) - ReLU - Dropout -
    Linear(4096, classes).  Layers are created in the script's order (conv, bn, conv, bn, ..., fc1-3), so
    the seeded initialisation and `parameters()` line up with the reference's."""

    def __init__(self, num_classes=10, device="hip", img_size=(32, 32)):
        super().__init__()
        cin = 3
        for si, widths in enumerate(_VGG16_STAGES, 1):
            for ci, cout in enumerate(widths, 1):
                setattr(self, f"conv{si}_{ci}", nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1,
                                                         bias=False, device=device))
                setattr(self, f"bn{si}_{ci}", nn.BatchNorm2d(cout, device=device))
                cin = cout
            setattr(self, f"pool{si}", nn.MaxPool2d(kernel_size=2, stride=2))
        side = img_size[0] // 2 ** len(_VGG16_STAGES)
        self.fc1 = nn.Linear(cin * side * side, 4096, device=device)
        self.fc2 = nn.Linear(4096, 4096, device=device)
        self.fc3 = nn.Linear(4096, num_classes, device=device)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)

    def forward(self, x):
        if not isinstance(x, Tensor):
            x = Tensor(x)
        for si, widths in enumerate(_VGG16_STAGES, 1):
            for ci in range(1, len(widths) + 1):
                x = self.relu(getattr(self, f"bn{si}_{ci}")(getattr(self, f"conv{si}_{ci}")(x)))
            x = getattr(self, f"pool{si}")(x)
        x = x.reshape(x.shape[0], -1)
        x = self.dropout(self.relu(self.fc1(x)))
        x = self.dropout(self.relu(self.fc2(x)))
        return self.fc3(x)


def make_divisible(channels, divisor=8):
    """test/MobileNet.py:37-43: round a scaled channel count to a multiple of `divisor`, never
    dropping more than 10 %."""
    rounded = int(channels + divisor / 2) // divisor * divisor
    out = divisor if rounded < divisor else rounded
    if out < 0.9 * channels:
        out += divisor
    return out


class ConvBlock(nn.Module):
    """test/MobileNet.py:18-35: 3x3 conv (no bias) -> BatchNorm2d -> maximum(x, 0)."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, device="hip"):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, kernel_size=kernel_size, stride=stride, padding=padding,
                              bias=False, device=device)
        self.bn = nn.BatchNorm2d(out_ch, device=device)

    def forward(self, x):
        return tensor.maximum(self.bn(self.conv(x)), 0)


class MobileNetV1(nn.Module):
    """test/MobileNet.py:48-108 ("simplified": plain conv blocks in place of the depthwise-separable
    ones): stem conv+BN+ReLU, eight ConvBlocks (strides 1,2,1,2,1,1,1,1), two `mean(axis=2)` calls
    (each divides by the total element count), Linear."""
    _BLOCKS = ((64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 1), (512, 1), (512, 1))

    def __init__(self, num_classes=10, width_multiplier=1.0, device="hip", img_size=(32, 32)):
        super().__init__()
        self.device, self.img_size = device, img_size

        def c(ch):
            return make_divisible(ch * width_multiplier)
        self.conv1 = nn.Conv2d(3, c(32), kernel_size=3, stride=1, padding=1, bias=False, device=device)
        self.bn1 = nn.BatchNorm2d(c(32), device=device)
        cin = c(32)
        for i, (ch, stride) in enumerate(self._BLOCKS, 1):
            setattr(self, f"block{i}", ConvBlock(cin, c(ch), stride=stride, device=device))
            cin = c(ch)
        self.last_channels = cin
        self.fc = nn.Linear(cin, num_classes, device=device)

    def forward(self, x):
        if not isinstance(x, Tensor):
            x = Tensor(x)
        x = tensor.maximum(self.bn1(self.conv1(x)), 0)
        for i in range(1, len(self._BLOCKS) + 1):
            x = getattr(self, f"block{i}")(x)
        x = tensor.mean(tensor.mean(x, axis=2), axis=2)
        return self.fc(x)


class PlainResidualBlock(ResidualBlock):
    """The block with the forward() the reference's script has, line for line (test/ResNet_CIFAR10_cuda.py:31-47; with
    `relu_after_add` test/ResNet.py:39-62): plain module calls, `out + identity`.  On the hip device the modules record
    their calls and launch the same fused kernels as ResidualBlock.forward (nn/lazy.py)."""

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out = self.bn1(out)
        out = self.conv2(out)
        out = self.bn2(out)
        if self.downsample is not None:
            for layer in self.downsample:
                identity = layer(identity)
        out = out + identity
        return nn.functional.relu(out) if self.relu_after_add else out


class PlainResNet(ResNet):
    """ResNet with the reference script's forward() (test/ResNet_CIFAR10_cuda.py:84-99): no private entry point, no
    restructuring -- what an unmodified user model looks like."""

    def forward(self, x):
        if not isinstance(x, Tensor):
            x = Tensor(x)
        x = self.conv1(x)
        x = self.bn1(x)
        x = self.relu(x)
        if self._stem_pool:
            x = self.pool1(x)
        for stage in (self.layer1, self.layer2, self.layer3):
            for block in stage:
                x = block(x)
                if not isinstance(x, Tensor):
                    x = Tensor(x)
        self.backward_milestone = x
        for block in self.layer4:
            x = block(x)
            if not isinstance(x, Tensor):
                x = Tensor(x)
        x = tensor.mean(x, axis=2)
        x = tensor.mean(x, axis=2)
        x = self.fc(x)
        return x


def ResNet18Plain(num_classes=10, device="hip", register_blocks=True):
    """ResNet-18 (CIFAR variant) written as the reference writes it; same parameters, same initialisation order and the same
    launches as ResNet18 on the hip device."""
    return PlainResNet(PlainResidualBlock, [2, 2, 2, 2], num_classes, device, register_blocks)


def late_and_early_parameters(model, params):
    """Split `params` (an optimizer's list) into those used only behind model.backward_milestone
    (layer4 and fc of a ResNet: their gradients are complete first) and the rest, keeping order."""
    late_ids = set()
    for m in list(model.layer4) + [model.fc]:
        stack = [m]
        while stack:
            mod = stack.pop()
            for p in mod._parameters.values():
                if p is not None:
                    late_ids.add(id(p))
            stack.extend(c for c in mod._modules.values() if c is not None)
            ds = getattr(mod, "downsample", None)
            if isinstance(ds, (list, tuple)):
                stack.extend(ds)
    late = [p for p in params if id(p) in late_ids]
    early = [p for p in params if id(p) not in late_ids]
    return late, early


def ResNet18(num_classes=10, device="hip", register_blocks=True):
    return ResNet(ResidualBlock, [2, 2, 2, 2], num_classes, device, register_blocks)


def ResNet18Wide(num_classes=10, device="hip", register_blocks=True):
    """test/ResNet.py:65-154: 64/128/256/512 channels on 32 x 32 inputs without a stem pool, ReLU after every residual
    add; 11 173 962 parameters, 852 GFLOP per training step at batch 256 (a stress shape, not the graded
    configuration).  The script keeps its blocks in Python lists as well (F6): register_blocks=False reproduces that."""
    return ResNet(ResidualBlock, [2, 2, 2, 2], num_classes, device, register_blocks, widths=(64, 128, 256, 512),
                  stem_pool=False, relu_after_add=True)


def all_block_parameters(model):
    """Every parameter of a ResNet including unregistered list-held blocks, in construction
    order (for checks on the register_blocks=False variant)."""
    seen, out = set(), []

    def visit(m):
        for p in m._parameters.values():
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        for c in m._modules.values():
            if c is not None:
                visit(c)
        for v in m.__dict__.values():
            if isinstance(v, (list, tuple)):
                for item in v:
                    if isinstance(item, nn.Module):
                        visit(item)
    visit(model)
    return sorted(out, key=lambda p: p.unique_id)


def smooth_one_hot(labels, num_classes=10, eps=0.0):
    """one-hot fp32 targets, optionally label-smoothed (test/ResNet_CIFAR10_cuda.py:181-183)."""
    y = np.eye(num_classes, dtype=np.float32)[np.asarray(labels)]
    if eps:
        y = y * (1 - eps) + eps / num_classes
    return y.astype(np.float32)


def train_step(model, criterion, optimizer, inputs, targets, sync=None):
    """One batch: forward, loss, zero_grad, backward, (gradient all-reduce), optimizer step."""
    outputs = model(inputs)
    loss = criterion(outputs, targets)
    optimizer.zero_grad()
    loss.backward()
    if sync is not None:
        sync()
    optimizer.step()
    return loss, outputs


def end_of_step():
    Graph.free_graph()
