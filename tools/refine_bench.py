"""Timing of the refinement-stage grid passes at the BASELINE grid size (160^3) on the GPU, next to the CPU oracle.
    gpurun -- python tools/refine_bench.py [side]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402
from voxe_hip.workload import refine_scene  # noqa: E402
from voxe_hip import ops  # noqa: E402

from oracle import voxe_oracle as vo  # noqa: E402


def timed(fn, reps=3):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return out, best * 1e3


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    dev = torch.device("cuda:0")
    dens, col, edit, pick = refine_scene(side)
    d_dev, c_dev = dens[..., 0].contiguous().to(dev), col.to(dev)
    (node, cap), ms_build = timed(lambda: ops.graph_build(d_dev, c_dev, 0.1, True))
    term = np.zeros((side,) * 3, np.int8)
    term[edit & (node.cpu().numpy() > 0)] = 1
    term[tuple(pick.T)] = -1
    t_dev = torch.from_numpy(term).to(dev)
    (seg, flow), ms_cut = timed(lambda: ops.graphcut(node, t_dev, cap), reps=2)
    mask = (d_dev > 0) | (torch.rand(d_dev.shape, device=dev) < 0.01)
    (labels, ncomp), ms_cc = timed(lambda: ops.cc_largest_k(mask, 10))
    print(f"grid {side}^3: nodes {int(node.sum())}, edit seeds {int((term > 0).sum())}, object seeds {int((term < 0).sum())}")
    print(f"HIP   graph_build {ms_build:8.2f} ms   graphcut {ms_cut:9.2f} ms (flow {flow})   cc_largest_k {ms_cc:7.2f} ms ({ncomp} components)")
    t0 = time.perf_counter()
    node_o, cap_o = vo.graph_build(dens[..., 0].numpy(), col.numpy(), 0.1, True)
    t1 = time.perf_counter()
    seg_o, flow_o, _ = vo.graphcut(node_o, term, cap_o)
    t2 = time.perf_counter()
    lab_o, n_o = vo.cc_largest_k(mask.cpu().numpy(), 10)
    t3 = time.perf_counter()
    print(f"oracle graph_build {(t1 - t0) * 1e3:8.2f} ms   graphcut {(t2 - t1) * 1e3:9.2f} ms (flow {flow_o})   cc_largest_k {(t3 - t2) * 1e3:7.2f} ms ({n_o} components)")
    print("labels equal:", bool(np.array_equal(seg.cpu().numpy(), seg_o)), bool(np.array_equal(labels.cpu().numpy(), lab_o)))


if __name__ == "__main__":
    main()
