"""A plain-torch restatement of the flat functions of the upstream checkouts' compiled extension (``dsgn._C`` / ``model._C``): TEST
INFRASTRUCTURE.  It plays the part of the CUDA build a user would have on an NVIDIA box - the stand-in checkouts' Python wrappers
(tests/fake_upstream/dsgn_checkout/dsgn/layers/) import ``dsgn._C`` exactly as upstream's do - so that the un-adopted stand-in network can
run (CPU or GPU, torch operators only) and be compared with the same network on libadvengine's shim (upstream_shims/ext_C.py).
Semantics: the published constructions, as oracle/oracle_np.py restates them (psv_build / psv_build_lerp / sigmoid_focal_loss / nms)."""
import torch

IS_TORCH_REFERENCE = True


def _planes(shift, b, w):
    s = shift if shift.dim() == 2 else shift.unsqueeze(0).expand(b, -1)
    for bi in range(b):
        for di in range(s.shape[1]):
            v = s[bi, di]
            if v.is_floating_point():
                sf = float(min(max(float(v), 0.0), float(w))) if float(v) == float(v) else 0.0
                sf = float(torch.tensor(sf, dtype=torch.float32))
                s0 = int(sf // 1)
                w1 = float(torch.tensor(sf, dtype=torch.float32) - torch.tensor(float(s0), dtype=torch.float32))
                yield bi, di, s0, (s0 + 1 if w1 > 0 else s0), 1.0 - w1, w1
            else:
                s0 = max(0, min(int(v), w))
                yield bi, di, s0, s0, 1.0, 0.0


def build_cost_volume_forward(left, right, shift):
    b, c, h, w = left.shape
    d = shift.shape[-1]
    cost = left.new_zeros((b, 2 * c, d, h, w))
    rz = torch.cat([right.new_zeros((b, c, h, w + 1)), right], dim=3)                  # rz[..., w + 1 + i] = right[i]
    for bi, di, s0, sc, w0, w1 in _planes(shift, b, w):
        if sc >= w:
            continue
        cost[bi, :c, di, :, sc:] = left[bi, :, :, sc:]
        lo = w + 1 + sc - s0
        a = rz[bi, :, :, lo:lo + (w - sc)]
        if w1 > 0:
            cost[bi, c:, di, :, sc:] = w0 * a + w1 * rz[bi, :, :, lo - 1:lo - 1 + (w - sc)]
        else:
            cost[bi, c:, di, :, sc:] = a
    return cost


def build_cost_volume_backward(grad_cost, shift):
    b, c2, d, h, w = grad_cost.shape
    c = c2 // 2
    gl, gr = grad_cost.new_zeros((b, c, h, w)), grad_cost.new_zeros((b, c, h, w))
    for bi, di, s0, sc, w0, w1 in _planes(shift, b, w):
        if sc >= w:
            continue
        gl[bi, :, :, sc:] += grad_cost[bi, :c, di, :, sc:]
        gm = grad_cost.new_zeros((c, h, 2 * w + 2))
        gm[:, :, sc:w] = grad_cost[bi, c:, di, :, sc:]
        gr[bi] += w0 * gm[:, :, s0:s0 + w] + (w1 * gm[:, :, s0 + 1:s0 + 1 + w] if w1 > 0 else 0)
    return gl, gr


def _focal_terms(logits, targets, gamma, alpha):
    x = logits
    t = targets.reshape(-1, 1).to(torch.int64)
    cls = torch.arange(1, x.shape[1] + 1, device=x.device).unsqueeze(0)
    p = torch.sigmoid(x)
    lp = torch.clamp(x, max=0) - torch.log1p(torch.exp(-x.abs()))
    lq = torch.clamp(-x, max=0) - torch.log1p(torch.exp(-x.abs()))
    return p, lp, lq, (t == cls), ((t >= 0) & (t != cls))


def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    p, lp, lq, pos, neg = _focal_terms(logits, targets, gamma, alpha)
    zero = torch.zeros_like(p)
    return torch.where(pos, -alpha * (1 - p) ** gamma * lp, zero) + torch.where(neg, -(1 - alpha) * p ** gamma * lq, zero)


def sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha):
    p, lp, lq, pos, neg = _focal_terms(logits, targets, gamma, alpha)
    zero = torch.zeros_like(p)
    g = torch.where(pos, alpha * (1 - p) ** gamma * (gamma * p * lp - (1 - p)), zero) + \
        torch.where(neg, (1 - alpha) * p ** gamma * (p - gamma * (1 - p) * lq), zero)
    return g * d_losses


def nms(dets, scores, threshold):
    """greedy suppression in descending-score order (stable for ties), legacy +1 areas, suppressed when IoU > threshold"""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    order = torch.sort(scores.reshape(-1), descending=True, stable=True)[1]
    b = dets[order, :4].float().cpu()
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    removed, keep = [False] * len(b), []
    for i in range(len(b)):
        if removed[i]:
            continue
        keep.append(i)
        for j in range(i + 1, len(b)):
            if removed[j]:
                continue
            ww = max(float(torch.minimum(b[i, 2], b[j, 2]) - torch.maximum(b[i, 0], b[j, 0]) + 1), 0.0)
            hh = max(float(torch.minimum(b[i, 3], b[j, 3]) - torch.maximum(b[i, 1], b[j, 1]) + 1), 0.0)
            inter = torch.tensor(ww, dtype=torch.float32) * torch.tensor(hh, dtype=torch.float32)
            if float(inter / (area[i] + area[j] - inter)) > float(torch.tensor(threshold, dtype=torch.float32)):
                removed[j] = True
    return order[torch.tensor(keep, dtype=torch.int64, device=order.device)]
