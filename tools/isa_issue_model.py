"""Static VALU-issue model of a kernel's hot loop: every VALU instruction of the compiler's assembly is priced with the cost
measured by tools/microbench/valu_rate.hip (profiles/r06_valu_rate.txt: clk per wave64 instruction per SIMD at >= 2 resident
waves) -- VERDICT r05 item 1(b): "histogram the backward loop's ISA by class x measured cost".

    python tools/isa_issue_model.py voxe_render_tile4.hip 'render_bwd_tile4_kernel<8, false, 0>' [--blocks] [--json out.json]

What the microbenchmark found (gfx950, MI355X):
  * 2.15 clk : v_fma / v_fmac / v_fmaak / v_fmamk / v_mul / v_add / v_sub / v_subrev _f32, v_add / v_sub / v_subrev _u32,
               v_and / v_or / v_xor _b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32 -- ONLY without an SGPR source operand, DPP
               or SDWA (v_fma_f32 with one SGPR source: 4.27);
  * 4.2 clk  : everything else that is not transcendental -- min / max / med3, compares, v_cndmask, floor / fract / trunc /
               rndne, every conversion, v_lshlrev_b32, v_lshl_add / v_add3 / v_and_or / v_bfe / v_perm, 24-bit and 32-bit integer
               multiplies, v_mad_u64_u32, every f64 add / mul / fma, packed f32, DPP, v_readlane / v_readfirstlane / v_writelane;
  * 8.1 clk  : v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos _f32;  16.1 clk: v_rcp / v_rsq / v_sqrt _f64;
  * 22.8 clk : v_cndmask_b32 (VOP2, implicit vcc) when vcc was last written by a SCALAR instruction (s_and_b64 vcc, ...)
               -- behind a v_cmp it is 4.0.
A single wave issues one VALU instruction per ~4.3 clk whatever its class: the 2.15 figure needs two or more resident waves.

The loop: blocks are weighted by how often they run per sample-loop iteration where that is knowable from the assembly alone
(everything inside the innermost loop that contains the 8 texel gathers counts once, blocks behind a conditional branch that
skips them count once as well: an upper bound of the per-iteration cost); `--blocks` prints the per-block table so the weights can
be audited.  bench.py combines the static mix with the DYNAMIC class counters of the PMC passes (SQ_INSTS_VALU_{TRANS_F32,
MUL_F64, ADD_F64, FMA_F64, CVT, ...}): see bench.py: issue_model()."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vox-e_amd", "voxe_hip"))
import build as b  # noqa: E402

FAST = {
    "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
TRANS64 = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}
COST = {"fast": 2.15, "slow": 4.2, "trans": 8.1, "trans64": 16.1, "cnd_salu_vcc": 22.8}
# PMC class of an opcode (what the SQ_INSTS_VALU_* counters count), for the dynamic re-weighting in bench.py
def pmc_class(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in TRANS:
        return "TRANS_F32"
    if base in TRANS64:
        return "TRANS_F64"
    if re.match(r"v_(fma|mad|fmac)_f64", base):
        return "FMA_F64"
    if re.match(r"v_mul_f64", base):
        return "MUL_F64"
    if re.match(r"v_add_f64", base):
        return "ADD_F64"
    if base.startswith("v_cvt_"):
        return "CVT"
    if re.match(r"v_(fma|fmac|fmaak|fmamk|mad|mac)_f32", base):
        return "FMA_F32"
    if re.match(r"v_mul_f32|v_mul_legacy_f32", base):
        return "MUL_F32"
    if re.match(r"v_(add|sub|subrev)_f32", base):
        return "ADD_F32"
    return "OTHER"


SGPR_RE = re.compile(r"\b(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|scc|ttmp\d+)\b")


def classify(op, operands, vcc_writer):
    base = re.sub(r"_(e32|e64)$", "", op)
    if base in TRANS:
        return "trans"
    if base in TRANS64:
        return "trans64"
    if base.endswith("_dpp") or base.endswith("_sdwa") or "quad_perm" in operands or "row_" in operands or "dst_sel" in operands:
        return "slow"
    if base == "v_cndmask_b32" and "_e64" not in op and vcc_writer == "S":
        return "cnd_salu_vcc"
    if base in FAST:
        srcs = operands.split(",")[1:]          # (the destination is never an SGPR for these)
        if any(SGPR_RE.search(s) for s in srcs):
            return "slow"
        return "fast"
    return "slow"


def assembly(src_file):
    extra = os.environ.get("ISA_EXTRA_FLAGS", "").split()          # (variants: -DVOXE_... experiments)
    out = f"/tmp/isa_issue_{os.path.basename(src_file)}{'_' + str(abs(hash(tuple(extra))) % 100000) if extra else ''}.s"
    src = os.path.join(b.CSRC, src_file)
    if not (os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(os.path.join(b.CSRC, f)) for f in os.listdir(b.CSRC))):
        subprocess.check_call([b.hipcc(), *b.FLAGS, *extra, "-I", b.INCLUDE, "-S", "--cuda-device-only", src, "-o", out],
                              stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_body(asm, demangled_filter):
    names = re.findall(r"^(_Z\w+):", asm, flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    want = demangled_filter.replace(" ", "")
    for mangled, d in zip(names, dem):
        if want in d.replace(" ", "").replace("voxe::", "") and ".kd" not in mangled:
            m = re.search(r"^" + re.escape(mangled) + r":.*?^\.Lfunc_end\d+:", asm, flags=re.M | re.S)   # (a kernel may hold several s_endpgm)
            if m:
                return d.split("(")[0], m.group(0)
    raise SystemExit(f"no kernel matches {demangled_filter!r}")


def blocks_of(body):
    blocks, cur = [], {"name": "entry", "ins": []}
    blocks.append(cur)
    for line in body.split("\n")[1:]:
        t = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = {"name": m.group(1), "ins": []}
            blocks.append(cur)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.startswith("//"):
            continue
        t = t.split(";")[0].strip()
        op = t.split()[0]
        cur["ins"].append((op, t[len(op):].strip()))
    return blocks


def price(blocks):
    """annotate every block with its VALU cost; vcc's last writer is tracked inside a block (unknown = VALU at block entry)"""
    for blk in blocks:
        writer = "V"
        cls_count, pmc_count = collections.Counter(), collections.Counter()
        other = collections.Counter()
        pmc_cost = collections.Counter()
        for op, operands in blk["ins"]:
            first = operands.split(",")[0].strip() if operands else ""
            if op.startswith("v_"):
                c = classify(op, operands, writer)
                cls_count[c] += 1
                pc = pmc_class(op)
                pmc_count[pc] += 1
                pmc_cost[pc] += COST[c]
                if first.startswith("vcc") or (op.startswith("v_cmp") and not first.startswith("s[")) or op.startswith("v_div_scale"):
                    writer = "V"
            else:
                if op.startswith("s_") and first.startswith("vcc"):
                    writer = "S"
                kind = ("ds" if op.startswith("ds_") else "vmem_atomic" if "atomic" in op else "vmem_load" if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))
                        else "vmem_store" if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")) else "smem" if op.startswith("s_load") or op.startswith("s_buffer_load")
                        else "branch" if op.startswith("s_cbranch") or op == "s_branch" else "waitcnt" if op.startswith("s_waitcnt") else "nop" if op == "s_nop" else "salu")
                other[kind] += 1
        blk["cls"], blk["pmc"], blk["pmc_cost"], blk["other"] = cls_count, pmc_count, pmc_cost, other
        blk["valu"] = sum(cls_count.values())
        blk["cost"] = sum(COST[c] * n for c, n in cls_count.items())
    return blocks


def loops_of(blocks):
    """innermost loops: (first block index, last block index) of every backward branch"""
    index = {blk["name"]: i for i, blk in enumerate(blocks)}
    loops = []
    for i, blk in enumerate(blocks):
        for op, operands in blk["ins"]:
            if (op.startswith("s_cbranch") or op == "s_branch") and operands in index and index[operands] <= i:
                loops.append((index[operands], i))
    return loops


def hot_loop(blocks, marker=lambda blk: blk["other"]["vmem_load"] >= 8 or blk["other"]["ds"] >= 32):
    """the smallest loops that contain a marker block (8 texel gathers / the 32 LDS adds of the deposit), one per march axis"""
    loops = loops_of(blocks)
    chosen = []
    has_deposit = any(blk["other"]["ds"] >= 32 for blk in blocks)
    for i, blk in enumerate(blocks):
        # the deposit block (unique per march instantiation of a backward); kernels without one (the forward): the gather block
        if (blk["other"]["ds"] >= 32) if has_deposit else (blk["other"]["vmem_load"] >= 8):
            cands = [lp for lp in loops if lp[0] <= i <= lp[1]]
            if cands:
                chosen.append(min(cands, key=lambda lp: lp[1] - lp[0]))
    return sorted(set(chosen))


def summarise(blocks, lo, hi):
    tot_cls, tot_pmc, tot_pmc_cost, tot_other = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    for blk in blocks[lo:hi + 1]:
        tot_cls.update(blk["cls"]); tot_pmc.update(blk["pmc"]); tot_pmc_cost.update(blk["pmc_cost"]); tot_other.update(blk["other"])
    valu = sum(tot_cls.values())
    cost = sum(COST[c] * n for c, n in tot_cls.items())
    return {"blocks": hi - lo + 1, "valu": valu, "cost_clk": round(cost, 1), "clk_per_valu": round(cost / max(valu, 1), 3),
            "by_cost_class": dict(tot_cls), "by_pmc_class": dict(tot_pmc),
            "clk_per_valu_by_pmc_class": {k: round(tot_pmc_cost[k] / tot_pmc[k], 3) for k in tot_pmc}, "other": dict(tot_other)}


def model(src_file, kernel_filter):
    name, body = kernel_body(assembly(src_file), kernel_filter)
    blocks = price(blocks_of(body))
    loops = hot_loop(blocks)
    out = {"kernel": name, "source": src_file, "cost_table_clk": COST, "loops": []}
    for lo, hi in loops:
        s = summarise(blocks, lo, hi)
        s["first_block"], s["last_block"] = blocks[lo]["name"], blocks[hi]["name"]
        out["loops"].append(s)
    whole = summarise(blocks, 0, len(blocks) - 1)
    out["whole_kernel"] = whole
    # the mix the dynamic model uses: the mean over the hot loops (the three march-axis instantiations are the same code)
    if out["loops"]:
        agg_cost, agg_n = collections.Counter(), collections.Counter()
        for lo, hi in loops:
            for blk in blocks[lo:hi + 1]:
                agg_cost.update(blk["pmc_cost"]); agg_n.update(blk["pmc"])
        out["clk_per_valu_by_pmc_class"] = {k: round(agg_cost[k] / agg_n[k], 3) for k in agg_n}
        out["clk_per_valu"] = round(sum(agg_cost.values()) / sum(agg_n.values()), 3)
        out["valu_share_by_pmc_class"] = {k: round(agg_n[k] / sum(agg_n.values()), 4) for k in agg_n}
    return out, blocks, loops


HEADLINE = [("voxe_render_tile4.hip", "render_bwd_tile4_kernel<8, false, 0>"), ("voxe_render_tile4.hip", "render_fwd_tile4w_kernel<false>")]


def occupancy_of(src_file):
    """waves per SIMD / VGPRs / LDS per kernel from the compiler's resource remarks (tools/kernel_resources.py)"""
    src = os.path.join(b.CSRC, src_file)
    out = subprocess.run([b.hipcc(), *b.FLAGS, "-I", b.INCLUDE, "-c", src, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                          "-o", "/dev/null"], capture_output=True, text=True).stderr
    names = subprocess.run(["c++filt"], input="\n".join(re.findall(r"Function Name: (\S+)", out)), capture_output=True, text=True).stdout.splitlines()
    res = {}
    for blk, name in zip(re.split(r"remark: .*?Function Name: ", out)[1:], names):
        g = lambda k: int(re.search(re.escape(k) + r": (\d+)", blk).group(1))  # noqa: E731
        res[name.split("(")[0].replace("void ", "")] = {"occupancy_waves_per_simd": g("Occupancy [waves/SIMD]"), "vgprs": g("VGPRs"),
                                                        "lds_bytes": g("LDS Size [bytes/block]"), "scratch_bytes_per_lane": g("ScratchSize [bytes/lane]")}
    return res


def write_profile(path):
    """profiles/r06_issue_model.json: the static mix of the headline kernels, stamped with the hash of the kernel sources"""
    kernels = {}
    for src_file, filt in HEADLINE:
        out, _, _ = model(src_file, filt)
        key = out["kernel"].replace("void ", "")
        occ = occupancy_of(src_file).get(key, {})
        kernels[key] = {"source": src_file, "clk_per_valu": out["clk_per_valu"], "clk_per_valu_by_pmc_class": out["clk_per_valu_by_pmc_class"],
                        "valu_share_by_pmc_class": out["valu_share_by_pmc_class"], "hot_loops": [
                            {k: lp[k] for k in ("first_block", "last_block", "valu", "cost_clk", "clk_per_valu", "by_cost_class", "other")} for lp in out["loops"]],
                        **occ}
    doc = {"source_hash": b.source_hash(), "cost_table_clk": COST, "cost_source": "profiles/r06_valu_rate.txt (tools/microbench/valu_rate.hip)",
           "how": "python tools/isa_issue_model.py --write : every VALU instruction of the compiler's assembly of the kernel's sample loop(s) priced "
                  "by class; bench.py multiplies the per-PMC-class means with the dynamic class counters of profiles/*_pmc_summary.json",
           "kernels": kernels}
    json.dump(doc, open(path, "w"), indent=1)
    print(path, {k: v["clk_per_valu"] for k, v in kernels.items()})


if __name__ == "__main__":
    if "--write" in sys.argv:
        write_profile(os.path.join(ROOT, "profiles", "r06_issue_model.json"))
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out, blocks, loops = model(args[0], args[1])
    if "--blocks" in sys.argv:
        for lo, hi in loops:
            print(f"loop {blocks[lo]['name']} .. {blocks[hi]['name']}")
            for blk in blocks[lo:hi + 1]:
                if blk["valu"] + sum(blk["other"].values()) >= 12:
                    print(f"  {blk['name']:<12} valu {blk['valu']:>4} cost {blk['cost']:>7.1f}  {dict(blk['cls'])}  {dict(blk['other'])}")
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print(json.dumps(out, indent=1))
