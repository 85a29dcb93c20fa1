"""Static report of the gfx950 machine code in libmldhip.so, per kernel: instruction mix and the code-generation traps this project
has met (DESIGN.md points 8, 27, 33, 34, 40) -- readable without a GPU.

  python tools/isa_report.py [--match den_loop] [--json out.json]

Columns: vgpr / scratch bytes / LDS bytes from the kernel descriptor notes; mfma, valu (v_* that are not matrix instructions),
ds_read / ds_write, global loads / stores, flat accesses (a ring pointer that lost its address space: point 34), ds_write_b16 (2-byte LDS
stores: the 8-way conflicted V^T staging of rounds 2-3), ds_read_b64_tr_b16, s_waitcnt vmcnt(0) (a drained memory counter: inside a
main loop it means a prefetch ring collapsed), s_cbranch (a kernel that gained branches around its loads: point 40)."""
import argparse, json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(ROOT, "motion-latent-diffusion_amd", "mld_hip", "libmldhip.so")


def report(lib=DEFAULT_LIB):
    tmp = tempfile.mkdtemp(prefix="mld_isa_")
    try:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "g.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "c.so")], check=True, capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-leading-addr", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        meta = {}
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1)) if re.search(r"\." + k + r":\s+(\d+)", blk) else None
            meta[name.group(1)] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"), lds_static=g("group_segment_fixed_size"),
                                       vgpr_spills=g("vgpr_spill_count"))
        out, cur = {}, None
        for ln in dis.splitlines():
            m = re.match(r"^<(.+)>:$", ln)
            if m:
                cur = m.group(1)
                out[cur] = dict(mfma=0, valu=0, ds_read=0, ds_write=0, ds_write_b16=0, ds_read_tr=0, ds_bpermute=0, global_load=0, global_store=0, flat=0,
                                vmcnt0=0, branches=0, barriers=0, insts=0, **meta.get(cur, {}))
                continue
            if not cur or not ln.startswith(("\t", " ")):
                continue
            t = ln.split("//")[0].strip()
            if not t:
                continue
            op = t.split()[0]
            k = out[cur]
            k["insts"] += 1
            if op.startswith("v_mfma"): k["mfma"] += 1
            elif op.startswith("v_"): k["valu"] += 1
            elif op.startswith("ds_read_b64_tr"): k["ds_read_tr"] += 1; k["ds_read"] += 1
            elif op.startswith("ds_bpermute"): k["ds_bpermute"] += 1
            elif op.startswith("ds_read"): k["ds_read"] += 1
            elif op.startswith("ds_write"):
                k["ds_write"] += 1
                if op.startswith("ds_write_b16"): k["ds_write_b16"] += 1
            elif op.startswith("global_load"): k["global_load"] += 1
            elif op.startswith("global_store"): k["global_store"] += 1
            elif op.startswith("flat_"): k["flat"] += 1
            elif op == "s_waitcnt" and "vmcnt(0)" in t: k["vmcnt0"] += 1
            elif op.startswith("s_cbranch"): k["branches"] += 1
            elif op == "s_barrier": k["barriers"] += 1
        filt = shutil.which("c++filt")
        if filt:
            dem = subprocess.run([filt], input="\n".join(out), capture_output=True, text=True).stdout.splitlines()
            out = {d: v for d, v in zip(dem, out.values())}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--match", default="")
    ap.add_argument("--json")
    a = ap.parse_args()
    rep = report(a.lib)
    if a.json:
        json.dump(rep, open(a.json, "w"), indent=1, sort_keys=True)
    cols = ("vgpr", "scratch", "mfma", "valu", "ds_read", "ds_write", "ds_write_b16", "ds_read_tr", "global_load", "global_store", "flat", "vmcnt0", "branches", "barriers")
    print(" ".join(f"{c:>7}" for c in cols), " kernel")
    for k, v in sorted(rep.items()):
        if a.match in k:
            print(" ".join(f"{str(v.get(c)):>7}" for c in cols), " ", k[:110])
