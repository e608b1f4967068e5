"""Image-comparison metrics of the parity bar: triangle-id coverage match, PSNR, and the
largest colour error on non-edge pixels (an edge pixel is one whose triangle id differs
from a 4-neighbour's in either image)."""
import numpy as np


def edge_mask(vis: np.ndarray) -> np.ndarray:
    t = vis >> 3  # triangle id without the clip-fan index
    e = np.zeros(t.shape, bool)
    e[:, 1:] |= t[:, 1:] != t[:, :-1]
    e[:, :-1] |= t[:, 1:] != t[:, :-1]
    e[1:, :] |= t[1:, :] != t[:-1, :]
    e[:-1, :] |= t[1:, :] != t[:-1, :]
    return e


def compare(vis_a, vis_b, col_a, col_b):
    ta, tb = vis_a >> 3, vis_b >> 3
    none_a, none_b = vis_a == 0xFFFFFFFF, vis_b == 0xFFFFFFFF
    same = (ta == tb) | (none_a & none_b)
    same &= none_a == none_b
    edges = edge_mask(vis_a) | edge_mask(vis_b)
    diff = np.abs(col_a.astype(np.int32) - col_b.astype(np.int32))
    mse = float(np.mean(diff[..., :3].astype(np.float64) ** 2))
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
    non_edge = ~edges & same
    max_non_edge = int(diff[non_edge].max()) if non_edge.any() else 0
    return dict(
        id_match=float(same.mean()),
        id_mismatch_off_edge=int((~same & ~edges).sum()),
        psnr=psnr,
        max_err_non_edge=max_non_edge,
        max_err=int(diff.max()),
        exact_pixels=float((diff.max(axis=2) == 0).mean()),
    )
