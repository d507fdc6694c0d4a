"""Registers / LDS / scratch / occupancy of the kernels of one source file, from the compiler's resource remarks.
    python tools/kernel_resources.py voxe_render_tile.hip [substring filter ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vox-e_amd", "voxe_hip"))
import build as b  # noqa: E402

src = os.path.join(b.CSRC, sys.argv[1])
filt = sys.argv[2:]
out = subprocess.run([b.hipcc(), *b.FLAGS, "-I", b.INCLUDE, "-c", src, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                      "-o", "/dev/null"], capture_output=True, text=True).stderr
names = subprocess.run(["c++filt"], input="\n".join(re.findall(r"Function Name: (\S+)", out)),
                       capture_output=True, text=True).stdout.splitlines()
for blk, name in zip(re.split(r"remark: .*?Function Name: ", out)[1:], names):
    name = name.split("(")[0].replace("voxe::", "")
    if filt and not all(f in name for f in filt):
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", blk).group(1)
    vals = [g("VGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")]
    print("%-78s VGPR %3s SGPR %3s scratch %4s occ %s LDS %s" % ((name[:78],) + tuple(vals)))
