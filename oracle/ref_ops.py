"""TEST INFRASTRUCTURE — direct NumPy restatements of the reference's hot-path functions, used
to check the fused HIP kernels at the BASELINE's full sizes (where driving the whole front-end
through the NumPy device would take minutes).  Each function follows the cited reference
lines; `tests/test_oracle_golden.py::test_ref_ops_match_goldens` pins them against the fixtures
generated from the reference itself.
"""
import numpy as np


def _im2col(xp, k, s):
    """(N,C,Hp,Wp) -> (N,C,k,k,oh,ow): nn/functional.py:263-283 (`col[:,:,i,j] = xp[:,:,i::s,j::s]`)."""
    N, C, Hp, Wp = xp.shape
    oh, ow = (Hp - k) // s + 1, (Wp - k) // s + 1
    col = np.empty((N, C, k, k, oh, ow), dtype=np.float32)
    for i in range(k):
        for j in range(k):
            col[:, :, i, j] = xp[:, :, i:i + oh * s:s, j:j + ow * s:s]
    return col


def conv2d_forward(x, w, bias, stride, pad):
    """nn/functional.py:316-344: pad -> im2col -> (N*oh*ow, C*k*k) @ (C*k*k, O) -> NCHW (+bias)."""
    N, C, H, W = x.shape
    O, _, k, _ = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    col = _im2col(xp, k, stride)
    oh, ow = col.shape[-2:]
    A = col.transpose(0, 4, 5, 1, 2, 3).reshape(N * oh * ow, C * k * k)
    y = (A @ w.reshape(O, -1).T).reshape(N, oh, ow, O).transpose(0, 3, 1, 2)
    if bias is not None:
        y = y + bias.reshape(1, O, 1, 1)
    return np.ascontiguousarray(y, dtype=np.float32)


def conv2d_wgrad(x, dy, k, stride, pad):
    """tensor.py:716: dB = A^T @ dC for the conv GEMM, reshaped to (O,C,k,k)."""
    N, C, H, W = x.shape
    O = dy.shape[1]
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    col = _im2col(xp, k, stride)
    oh, ow = col.shape[-2:]
    A = col.transpose(0, 4, 5, 1, 2, 3).reshape(N * oh * ow, C * k * k)
    dC = dy.transpose(0, 2, 3, 1).reshape(N * oh * ow, O)
    return np.ascontiguousarray((A.T @ dC).T.reshape(O, C, k, k), dtype=np.float32)


def conv2d_dgrad(dy, w, x_shape, stride, pad, mode="overwrite"):
    """tensor.py:708 (dA = dC @ B^T) -> __im2col2d.grad_fn (functional.py:285-294: k*k strided
    ASSIGNMENTS in (i,j) row-major order: last writer wins) -> __pad2d.grad_fn (:307-313).
    mode='accumulate' sums the overlapping windows instead (the true gradient)."""
    N, C, H, W = x_shape
    O, _, k, _ = w.shape
    oh, ow = dy.shape[-2:]
    dC = dy.transpose(0, 2, 3, 1).reshape(N * oh * ow, O)
    gcol = (dC @ w.reshape(O, -1)).reshape(N, oh, ow, C, k, k).transpose(0, 3, 4, 5, 1, 2)
    gx = np.zeros((N, C, H + 2 * pad, W + 2 * pad), dtype=np.float32)
    for i in range(k):
        for j in range(k):
            if mode == "overwrite":
                gx[:, :, i:i + oh * stride:stride, j:j + ow * stride:stride] = gcol[:, :, i, j]
            else:
                gx[:, :, i:i + oh * stride:stride, j:j + ow * stride:stride] += gcol[:, :, i, j]
    if pad:
        gx = gx[:, :, pad:-pad, pad:-pad]
    return np.ascontiguousarray(gx, dtype=np.float32)


def batchnorm2d_train(x, gamma, beta, eps=1e-5):
    """nn/modules/batchnorm.py:30-55 (biased variance)."""
    m = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.sum(axis=(0, 2, 3), keepdims=True, dtype=np.float32) / np.float32(m)
    diff = x - mean
    var = (diff * diff).sum(axis=(0, 2, 3), keepdims=True, dtype=np.float32) / np.float32(m)
    y = diff / (var + np.float32(eps)) ** np.float32(0.5) * gamma + beta
    return y.astype(np.float32), mean.astype(np.float32), var.astype(np.float32)


def batchnorm2d_backward(dy, x, gamma, mean, var, eps=1e-5):
    """Gradient of that expression graph as the reference's autograd produces it."""
    m = np.float32(x.shape[0] * x.shape[2] * x.shape[3])
    sd = (var + np.float32(eps)) ** np.float32(0.5)
    xhat = (x - mean) / sd
    dbeta = dy.sum(axis=(0, 2, 3), keepdims=True, dtype=np.float32)
    dgamma = (dy * xhat).sum(axis=(0, 2, 3), keepdims=True, dtype=np.float32)
    dx = (dy - dbeta / m - xhat * dgamma / m) * gamma / sd
    return dx.astype(np.float32), dgamma.astype(np.float32), dbeta.astype(np.float32)


def maxpool2d_forward(x, k):
    N, C, H, W = x.shape
    oh, ow = H // k, W // k
    return x[:, :, :oh * k, :ow * k].reshape(N, C, oh, k, ow, k).max(axis=(3, 5))


def maxpool2d_backward(x, y, dy, k):
    """tensor.max.grad_fn (tensor.py:779-791): (y == x) * grad for every tied maximum."""
    N, C, H, W = x.shape
    oh, ow = y.shape[-2:]
    dx = np.zeros_like(x)
    up = lambda a: np.repeat(np.repeat(a, k, axis=2), k, axis=3)
    dx[:, :, :oh * k, :ow * k] = (up(y) == x[:, :, :oh * k, :ow * k]) * up(dy)
    return dx


def cross_entropy(logits, target):
    """nn/functional.py:104-115, reduction='mean'."""
    m = logits.max(axis=1, keepdims=True)
    u = logits - m
    lse = np.log(np.exp(u).sum(axis=1, keepdims=True))
    return np.float32((-(u - lse) * target).sum() * np.float32(1.0 / logits.shape[0]))


def adam_step(p, g, v, s, t, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """optim/adam.py:28-64, fp32 arithmetic in the reference's order."""
    f = np.float32
    if wd > 0:
        g = g + p * f(wd)
    v = v * f(b1) + g * f(1 - b1)
    s = s * f(b2) + g ** f(2) * f(1 - b2)
    vh = v / f(1 - b1 ** t)
    sh = s / f(1 - b2 ** t)
    p = p + (vh / (sh ** f(0.5) + f(eps)) * f(lr)) * f(-1)
    return p.astype(np.float32), v.astype(np.float32), s.astype(np.float32)


def augment_batch(inputs, dec, pad=4):
    """test/ResNet_CIFAR10_cuda.py:129-148 with the random decisions passed in
    (dec[n] = ys, xs, flip, erase_y, erase_x, erase_h, erase_w, 0): reflect pad, crop, flip, erase, clip."""
    bs, c, h, w = inputs.shape
    padded = np.pad(inputs, ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode='reflect')
    out = np.empty_like(inputs)
    for i in range(bs):
        ys, xs = int(dec[i, 0]), int(dec[i, 1])
        out[i] = padded[i, :, ys:ys + h, xs:xs + w]
        if dec[i, 2]:
            out[i] = out[i][:, :, ::-1]
        if dec[i, 5] > 0:
            out[i, :, dec[i, 3]:dec[i, 3] + dec[i, 5], dec[i, 4]:dec[i, 4] + dec[i, 6]] = 0.0
    return np.clip(out, -1.0, 1.0)


def smooth_onehot(labels, num_classes, eps):
    """test/ResNet_CIFAR10_cuda.py:180-184."""
    onehot = np.eye(num_classes, dtype=np.float32)[np.asarray(labels).reshape(-1)]
    return (onehot * (1 - eps) + eps / num_classes).astype(np.float32)
