"""bf/utils/box_utils.py on libssdk (csrc/boxes.hip): to_corners, to_centroids, area, intersection, iou, generalized_iou, nms -- the
reference's names, argument order and return layouts (bf/utils/box_utils.py:16-194), computed by HIP kernels that round op for op like the
reference's separate torch ops.  The training / inference path never calls these (the IoU is fused into ``ssdk_encode_ground_truth``,
GIoU into the loss, NMS into ``ssdk_postprocess``); they exist for callers of the module itself, e.g. ``iou`` in front of
``matcher.match_per_prediction``.

Inputs: CUDA tensors are used where they are; CPU tensors and numpy arrays (the reference's ``to_torch`` decorator, :8-15) are copied to
the current CUDA device and the result comes back where the first argument lived.  There is no CPU implementation."""
import numpy as np
import torch

from ... import _lib


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('bf.utils.box_utils runs on libssdk (HIP): no GPU is available and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


class _Home(object):
    """Where the first argument lived: results go back there (numpy -> numpy, CPU tensor -> CPU tensor)."""

    def __init__(self, x):
        self.numpy = isinstance(x, np.ndarray)
        self.device = None if self.numpy else x.device

    def back(self, t):
        if self.numpy:
            return t.cpu().numpy()
        return t if t.device == self.device else t.to(self.device)


def _dev(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if x.device.type != 'cuda':
        x = x.to(_device())
    return x.float().contiguous()


def _boxes(x, what):
    t = _dev(x)
    if t.dim() < 1 or t.shape[-1] != 4:
        raise ValueError(f'{what}: boxes must be [..., 4], got {tuple(t.shape)}')
    return t


def to_corners(box):
    """[..., 4] (cx, cy, w, h) -> (x1, y1, x2, y2)   (box_utils.py:16-23)"""
    home, b = _Home(box), _boxes(box, 'to_corners')
    out = torch.empty_like(b)
    _lib.check(_lib.lib().ssdk_box_to_corners(_lib.ptr(b), _lib.ptr(out), b.numel() // 4, _lib.current_stream()), 'ssdk_box_to_corners')
    return home.back(out)


def to_centroids(box, inplace=False):
    """[..., 4] corners -> (cx, cy, w, h)   (box_utils.py:25-36).  ``inplace=True`` mutates ``box`` and returns None like the reference --
    and rounds the centre like its in-place branch (wh = max - min; c = min + wh / 2), not like the out-of-place one ((max + min) / 2)."""
    if inplace:
        if isinstance(box, np.ndarray) or box.device.type != 'cuda' or box.dtype != torch.float32 or not box.is_contiguous():
            res = to_centroids_form(box, True)
            if isinstance(box, np.ndarray):
                box[...] = res
            else:
                box.copy_(res)
            return None
        _lib.check(_lib.lib().ssdk_box_to_centroids(_lib.ptr(box), _lib.ptr(box), box.numel() // 4, 1, _lib.current_stream()), 'ssdk_box_to_centroids')
        return None
    return to_centroids_form(box, False)


def to_centroids_form(box, inplace_form):
    home, b = _Home(box), _boxes(box, 'to_centroids')
    out = torch.empty_like(b)
    _lib.check(_lib.lib().ssdk_box_to_centroids(_lib.ptr(b), _lib.ptr(out), b.numel() // 4, 1 if inplace_form else 0, _lib.current_stream()),
               'ssdk_box_to_centroids')
    return home.back(out)


def area(box):
    """[..., 4] corners -> [...]: clamp(x2 - x1, 0) * clamp(y2 - y1, 0)   (box_utils.py:38-46)"""
    home, b = _Home(box), _boxes(box, 'area')
    out = torch.empty(b.shape[:-1], dtype=torch.float32, device=b.device)
    _lib.check(_lib.lib().ssdk_box_area(_lib.ptr(b), _lib.ptr(out), b.numel() // 4, _lib.current_stream()), 'ssdk_box_area')
    return home.back(out)


def _pair(a, b, cartesian, what):
    A, B = _boxes(a, what), _boxes(b, what)
    if A.dim() != 2 or B.dim() != 2:
        raise ValueError(f'{what}: a and b must be [Boxes, 4]')
    if not cartesian:
        assert A.size() == B.size()   # box_utils.py:70
    return A, B


def intersection(a, b, cartesian=True, zero_incorrect=False):
    """[BoxesA, 4], [BoxesB, 4] -> [BoxesA, BoxesB, 4] (cartesian) or [Boxes, 4]: the corner box of the overlap   (box_utils.py:49-80)"""
    home = _Home(a)
    A, B = _pair(a, b, cartesian, 'intersection')
    out = torch.empty(((A.size(0), B.size(0), 4) if cartesian else (A.size(0), 4)), dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().ssdk_box_intersection(_lib.ptr(A), A.size(0), _lib.ptr(B), B.size(0), int(bool(cartesian)), int(bool(zero_incorrect)),
                                                _lib.ptr(out), _lib.current_stream()), 'ssdk_box_intersection')
    return home.back(out)


def _iou(a, b, cartesian, generalized, what):
    home = _Home(a)
    A, B = _pair(a, b, cartesian, what)
    out = torch.empty(((A.size(0), B.size(0)) if cartesian else (A.size(0),)), dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().ssdk_box_iou(_lib.ptr(A), A.size(0), _lib.ptr(B), B.size(0), int(bool(cartesian)), int(generalized), _lib.ptr(out),
                                       _lib.current_stream()), 'ssdk_box_iou')
    return home.back(out)


def iou(a, b, cartesian=True):
    """[BoxesA, 4], [BoxesB, 4] corners -> [BoxesA, BoxesB] (or [Boxes] when not cartesian)   (box_utils.py:83-101).  No +1, no epsilon:
    two degenerate boxes give NaN like the reference."""
    return _iou(a, b, cartesian, 0, 'iou')


def generalized_iou(a, b, cartesian=True):
    """https://arxiv.org/pdf/1902.09630v2.pdf   (box_utils.py:104-143)"""
    return _iou(a, b, cartesian, 1, 'generalized_iou')


def nms(boxes, scores, overlap_threshold, score_threshold, max_per_class=None, soft=False, sigma=0.5):
    """One class of one image (box_utils.py:166-194; the batched form the Postprocessor runs is ``ssdk_postprocess``).
    Returns ((boxes_picked [P, 4], scores_picked [P]), indexes_picked int64 [P]).  With ``max_per_class < Boxes`` the reference first takes
    ``topk(max_per_class, sorted=False)`` (:186-188) and ``indexes_picked`` index THAT subset, whose order it leaves to the library: here
    the subset is in descending score order (ties by ascending index).  Hard NMS follows torchvision.ops.nms's documented contract (:193;
    parity unpinned: torchvision is not part of the reference tree), soft NMS is the reference's own ``_soft_nms`` (:145-163)."""
    home = _Home(boxes)
    B, S = _boxes(boxes, 'nms'), _dev(scores)
    if B.dim() != 2 or S.dim() != 1 or S.size(0) != B.size(0):
        raise ValueError(f'nms: boxes [Boxes, 4] and scores [Boxes], got {tuple(B.shape)} and {tuple(S.shape)}')
    n = B.size(0)
    lib = _lib.lib()
    cap = 0 if max_per_class is None else int(max_per_class)
    if max_per_class is not None and cap <= 0:
        raise ValueError(f'nms: max_per_class={max_per_class}')
    k = n if (cap == 0 or cap >= n) else cap
    picked = torch.empty((max(k, 1),), dtype=torch.int64, device=B.device)
    pboxes = torch.empty((max(k, 1), 4), dtype=torch.float32, device=B.device)
    pscores = torch.empty((max(k, 1),), dtype=torch.float32, device=B.device)
    count = torch.zeros((1,), dtype=torch.int32, device=B.device)
    ws = _lib.scratch(lib.ssdk_nms_workspace_bytes(n), B.device, 'nms')
    _lib.check(lib.ssdk_nms(_lib.ptr(B), _lib.ptr(S), n, float(overlap_threshold), float(score_threshold), cap, int(bool(soft)), float(sigma),
                            _lib.ptr(picked), _lib.ptr(pboxes), _lib.ptr(pscores), _lib.ptr(count), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               'ssdk_nms')
    p = int(count.item())   # (the reference's result is ragged: one host read, like its own .nonzero())
    return (home.back(pboxes[:p]), home.back(pscores[:p])), home.back(picked[:p])
