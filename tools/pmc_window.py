#!/usr/bin/env python3
"""rocprofv3 --pmc passes of one bench.py workload -> profiles/pmc_<workload>.json.

The record is VERSIONED: it carries the commit it was collected at (`--head`, handed in by the caller - a GPU box has no
.git), `gpupoly_version()` and, per kernel, the hash of the kernel's ISA text in the library that ran
(mxx_amd/codeobj.py).  bench.py uses a kernel's counts only while that hash equals the hash of the kernel in the library it
is running with, and says `counters_stale` otherwise.

bench.py launches `gpupoly_marker_kernel` on each side of its timed region (ids 1 and 2).  Only the dispatches BETWEEN the
two markers are counted, so the set-up's launches of the same kernels (other sizes: trapdoor generation, operand sampling,
warm-up) stay out of the per-launch and per-step figures.  Counters of separate passes (FETCH_SIZE,
WRITE_SIZE, SQ_*) are joined per kernel; every pass runs the same command, so the window holds the same launches.

    pmc_window.py --workload m3a --steps 3 --head <commit> --out profiles/pmc_m3a.json PASSDIR [PASSDIR ...]

FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled (gfx950 reports half of the bytes of a wide streaming
read, MI355X_MICROARCH.md section HBM); WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_base(name: str) -> str:
    s = name.strip().strip("()").strip()
    if s.startswith("void "):
        s = s[5:]
    for ch in "<(":
        k = s.find(ch)
        if k > 0:
            s = s[:k]
    return s.strip()


def window_rows(passdir):
    rows = []
    for f in glob.glob(os.path.join(passdir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        return None
    # one row per (dispatch, counter)
    by_dispatch = collections.OrderedDict()
    for r in sorted(rows, key=lambda r_: int(r_["Dispatch_Id"])):
        d = by_dispatch.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = list(by_dispatch)
    marks = [i for i in ids if "gpupoly_marker_kernel" in by_dispatch[i]["name"]]
    if len(marks) < 2:
        raise SystemExit(f"{passdir}: fewer than two region markers in the dispatch list")
    lo, hi = marks[0], marks[1]  # the first timed region of the run
    return [by_dispatch[i] for i in ids if lo < i < hi]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--steps", type=int, required=True, help="--steps of the profiled bench.py command")
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default=None)
    ap.add_argument("--head", default=os.environ.get("MXX_HEAD", "unknown"), help="commit of the tree the passes ran on")
    ap.add_argument("passes", nargs="+")
    a = ap.parse_args()
    kern = collections.OrderedDict()
    for p in a.passes:
        win = window_rows(p)
        if win is None:
            continue
        seen = collections.Counter()
        for d in win:
            base = kernel_base(d["name"])
            k = kern.setdefault(base, {"launches": {}, "sum": collections.defaultdict(float), "full_name": d["name"][:160]})
            seen[base] += 1
            for c, v in d["c"].items():
                k["sum"][c] += v
        for base, cnt in seen.items():
            kern[base]["launches"][p] = cnt
    from mxx_amd import codeobj

    hashes = codeobj.kernel_isa_hashes()
    try:
        from mxx_amd import _ffi

        version = _ffi.lib().gpupoly_version().decode()
    except Exception as e:  # noqa: BLE001 - the record is still usable without the string
        version = f"unavailable: {e}"
    out = {"workload": a.workload, "steps": a.steps, "head": a.head, "gpupoly_version": version, "isa_hash_all": hashes.get("*"),
           "source": (a.command or f"rocprofv3 --pmc <counters> -- python3 bench.py --workload {a.workload} --steps {a.steps} "
                      "--warmup 1 --repeats 0 --no-cpu-baseline --no-trace") +
                     "; separate passes for FETCH_SIZE, WRITE_SIZE and the SQ counters; only dispatches between bench.py's region "
                     "markers counted (tools/pmc_window.py); FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950",
           "kernels": {}}
    total_bytes = 0.0
    have_traffic = False
    for base, k in kern.items():
        launches = max(k["launches"].values())
        rec = {"launches_per_step": launches / a.steps, "full_name": k["full_name"], "isa_hash": hashes.get(base)}
        for c, v in sorted(k["sum"].items()):
            rec[c] = v / a.steps  # per step, summed over the step's launches of this kernel
        if "FETCH_SIZE" in k["sum"] and "WRITE_SIZE" in k["sum"]:
            hbm = (2.0 * k["sum"]["FETCH_SIZE"] + k["sum"]["WRITE_SIZE"]) * 1024.0
            rec["hbm_bytes_per_step"] = hbm / a.steps
            rec["hbm_bytes_per_launch"] = hbm / launches
            total_bytes += hbm / a.steps
            have_traffic = True
        if k["sum"].get("SQ_THREAD_CYCLES_VALU") and k["sum"].get("SQ_ACTIVE_INST_VALU"):
            # active lanes per executed VALU instruction (the derived counter VALUUtilization of counter_defs.yaml, as a fraction)
            rec["lane_utilisation"] = round(k["sum"]["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["sum"]["SQ_ACTIVE_INST_VALU"]), 4)
        out["kernels"][base] = rec
    out["hbm_bytes_per_step"] = total_bytes if have_traffic else None
    json.dump(out, open(a.out, "w"), indent=1)
    print(f"{a.out}: {len(out['kernels'])} kernels, {sum(r['launches_per_step'] for r in out['kernels'].values()):.1f} launches per step")
    for base, r in out["kernels"].items():
        print(f"  {base:44s} x{r['launches_per_step']:<6.2f} VALU {r.get('SQ_INSTS_VALU', 0):14.0f}  HBM {r.get('hbm_bytes_per_step', 0) / 1e6:10.2f} MB per step")


if __name__ == "__main__":
    main()
