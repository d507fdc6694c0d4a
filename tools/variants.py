"""Build kernel variants of libvoxe_hip.so for A/B runs on the GPU: only voxe_render_tile.hip is recompiled with the
given -D flags, the other objects are reused.   python tools/variants.py tag "-DVOXE_TILE_ROT=13" ...
-> variants/libvoxe_hip_<tag>.so ; run with VOXE_HIP_LIB=variants/libvoxe_hip_<tag>.so python bench.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vox-e_amd", "voxe_hip"))
import build as b  # noqa: E402


def main():
    tag, flags = sys.argv[1], sys.argv[2:]
    b.build()
    out_dir = os.path.join(ROOT, "variants")
    os.makedirs(out_dir, exist_ok=True)
    which = os.environ.get("VARIANT_SRC", "voxe_render_tile.hip")   # one source file, or "all"
    rebuilt = b.SOURCES if which == "all" else [which]
    objs = []
    for name in b.SOURCES:
        if name in rebuilt:
            obj = os.path.join(out_dir, f"{name[:-4]}_{tag}.o")
            subprocess.check_call([b.hipcc(), *b.FLAGS, *flags, "-I", b.INCLUDE, "-c", os.path.join(b.CSRC, name), "-o", obj])
        else:
            obj = os.path.join(b.OBJ_DIR, name.replace(".hip", ".o"))
        objs.append(obj)
    lib = os.path.join(out_dir, f"libvoxe_hip_{tag}.so")
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    print(lib)


if __name__ == "__main__":
    main()
