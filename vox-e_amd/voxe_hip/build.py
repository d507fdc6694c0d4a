"""Compile the HIP sources in vox-e_amd/csrc into vox-e_amd/voxe_hip/libvoxe_hip.so (in-tree).

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build container; the
.so travels to the GPU box with the repository snapshot.

    python vox-e_amd/voxe_hip/build.py [--force] [--verbose]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libvoxe_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")

SOURCES = ["voxe_render.hip", "voxe_render_tile.hip", "voxe_render_tile4.hip", "voxe_render_tilew.hip", "voxe_render_scatter.hip", "voxe_render_region.hip", "voxe_grid_ops.hip", "voxe_refine.hip", "voxe_api.hip"]
HEADERS = ["voxe_device.hpp", "voxe_launch.hpp", "voxe_render_common.hpp", "voxe_tile_window.hpp"]

# -ffp-contract=off : the voxel-index arithmetic must round like the reference (no implicit FMA)
# -munsafe-fp-atomics: float atomicAdd -> global_atomic_add_f32 (no CAS loop)
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
    "-Wall", "-Wno-unused-function",
]


def source_hash() -> str:
    """identity of the kernel sources: sha256 over vox-e_amd/csrc/* and include/voxe.h (names + contents), 16 hex digits.
    Stamped into profiles/*_pmc_summary.json by tools/pmc_to_json.py and re-computed by bench.py, so that counter values
    are only ever combined with timings of the kernels they were collected on."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    files.append(os.path.join(INCLUDE, "voxe.h"))
    for path in files:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "voxe.h"), os.path.abspath(__file__)]
    # the library is newer than every source it is made of: nothing to do, whether or not the object files are at hand (they do
    # not travel to the GPU box: .gpurunignore)
    if not force and not extra_flags and _newer(LIB, [os.path.join(CSRC, s) for s in SOURCES] + hdrs):
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    todo = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not _newer(op, [sp] + hdrs):
            todo.append([hipcc(), *FLAGS, *extra_flags, "-I", INCLUDE, "-c", sp, "-o", op])
    rebuilt = bool(todo)
    if todo:
        # the translation units are independent: compile them side by side (the tile kernels alone take ~50 s)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as pool:
            list(pool.map(run, todo))
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
