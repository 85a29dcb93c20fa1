"""Machine-code identity of every gfx950 kernel in libmldhip.so: {demangled-ish symbol: sha256 of its disassembly}.

  python tools/kernel_hashes.py --out /tmp/a.json          # dump
  python tools/kernel_hashes.py --diff /tmp/a.json         # which kernels of the current build differ from a dump

Use: a source edit that is meant to ADD a variant (a new template argument, an option that is off by default) must leave the
machine code of the kernels the defaults launch untouched; when no GPU is at hand this is the proof that it did.  Same extraction
as bench.kernel_code_hash (llvm-objcopy -> clang-offload-bundler -> llvm-objdump), one pass over the code object."""
import argparse, hashlib, json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
LLVM = "/opt/rocm/lib/llvm/bin"


def dump(lib):
    tmp = tempfile.mkdtemp(prefix="mld_kh_")
    try:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "g.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "c.so")], check=True, capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-leading-addr", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        out, cur, body = {}, None, []
        for ln in dis.splitlines():
            m = re.match(r"^<(.+)>:$", ln)
            if m:
                if cur:
                    out[cur] = hashlib.sha256("\n".join(body).encode()).hexdigest()[:16]
                cur, body = m.group(1), []
            elif cur and ln.startswith(("\t", " ")):
                t = ln.split("//")[0].strip()
                if t:
                    body.append(t)
        if cur:
            out[cur] = hashlib.sha256("\n".join(body).encode()).hexdigest()[:16]
        filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
        if not filt:
            return out                                  # mangled names
        dem = subprocess.run([filt], input="\n".join(out), capture_output=True, text=True).stdout.splitlines()
        return {d: h for d, h in zip(dem, out.values())}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "motion-latent-diffusion_amd", "mld_hip", "libmldhip.so"))
    ap.add_argument("--out")
    ap.add_argument("--diff")
    a = ap.parse_args()
    cur = dump(a.lib)
    if a.out:
        json.dump(cur, open(a.out, "w"), indent=1, sort_keys=True)
        print(len(cur), "kernels ->", a.out)
    if a.diff:
        old = json.load(open(a.diff))
        ch = sorted(k for k in cur if k in old and old[k] != cur[k])
        print("changed:", len(ch), "new:", len(set(cur) - set(old)), "gone:", len(set(old) - set(cur)))
        for k in ch:
            print("  changed ", k[:150])
        for k in sorted(set(cur) - set(old)):
            print("  new     ", k[:150])
        for k in sorted(set(old) - set(cur)):
            print("  gone    ", k[:150])
