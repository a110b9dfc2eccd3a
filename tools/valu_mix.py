#!/usr/bin/env python3
"""Static VALU instruction mix of every kernel of libgpupoly -> profiles/valu_mix.json (versioned: every kernel's record
carries the hash of its ISA text, mxx_amd/codeobj.py; bench.py ignores a record whose hash is not today's).

bench.py's composed roofline prices a kernel's counted SQ_INSTS_VALU at `cycles_per_inst`: the mix-weighted issue cost
of its ISA at the measured per-instruction rates (profiles/r02_valu_rates.txt, 4 waves per SIMD: add / sub / and / or / xor
and the f32 add / mul class 2.5 cycles per wave64 instruction, everything else - integer multiplies, v_mad_u64_u32, shifts,
v_alignbit, min / max, compares + cndmask, conversions, every f64 op - 4.5).  The mix is STATIC (instructions in the
kernel's text, cold paths included), the count it is applied to is dynamic; for loop-dominated kernels the two agree to a
few percent.  Runs on the CPU: extracts the gfx950 code objects from mxx_amd/csrc/*.o into a temp dir and disassembles them.
"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
CHEAP = re.compile(r"^v_(add|sub|subrev|and|or|xor|xnor|not|mov)_(u32|i32|b32|co_u32|nc_u32)|^v_(addc|subb|subbrev)_co_u32|^v_(add|sub|mul)_f32|^v_add3_u32|^v_(and_or|or3|xad|xor3)_")
CHEAP_CYC, OTHER_CYC = 2.5, 4.5


def kernel_base(name):
    s = name.strip()
    if s.startswith("void "):
        s = s[5:]
    for ch in "<(":
        k = s.find(ch)
        if k > 0:
            s = s[:k]
    return s.strip()


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "valu_mix.json")
    tmp = tempfile.mkdtemp(prefix="valu_mix_")
    agg = collections.defaultdict(lambda: [0, 0])
    try:
        for obj in sorted(os.listdir(os.path.join(ROOT, "mxx_amd", "csrc"))):
            if not obj.endswith(".o"):
                continue
            local = os.path.join(tmp, obj)
            shutil.copy(os.path.join(ROOT, "mxx_amd", "csrc", obj), local)
            subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            co = [f for f in os.listdir(tmp) if f.startswith(obj + ".") and "gfx950" in f]
            if not co:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", os.path.join(tmp, co[0])], capture_output=True, text=True).stdout
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
                if m:
                    cur = kernel_base(m.group(1))
                    continue
                parts = ln.split()
                if cur is None or not parts:
                    continue
                op = parts[0]
                if not op.startswith("v_") or op.startswith("v_mfma") or op.startswith("v_accvgpr") or op.startswith("v_readlane") or \
                        op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
                    continue
                op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
                agg[cur][0] += 1
                if CHEAP.match(op):
                    agg[cur][1] += 1
            for f in os.listdir(tmp):
                os.remove(os.path.join(tmp, f))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sys.path.insert(0, ROOT)
    from mxx_amd import codeobj

    hashes = codeobj.kernel_isa_hashes()
    kernels = {}
    for base, (n, cheap) in sorted(agg.items()):
        if n < 8 or "kernel" not in base:
            continue
        kernels[base] = {"valu_static": n, "cheap_static": cheap, "isa_hash": hashes.get(base),
                         "cycles_per_inst": round((cheap * CHEAP_CYC + (n - cheap) * OTHER_CYC) / n, 3)}
    json.dump({"prices": {"cheap (add/sub/and/or/xor/mov, f32 add/mul)": CHEAP_CYC, "other": OTHER_CYC,
                          "source": "profiles/r02_valu_rates.txt (tools/valu_rates.hip, 4 waves per SIMD)"},
               "isa_hash_all": hashes.get("*"),
               "note": "static mix over the kernel's text (all template instances of a kernel pooled); applied to dynamic SQ_INSTS_VALU counts",
               "kernels": kernels}, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:48s} {v['valu_static']:8d} VALU, {100.0 * v['cheap_static'] / v['valu_static']:5.1f} % cheap -> {v['cycles_per_inst']} cycles")


if __name__ == "__main__":
    main()
