#!/usr/bin/env python3
"""Static VALU / SALU / LDS / VMEM instruction counts of one kernel per source line, from `hipcc -S -gline-tables-only
--cuda-device-only` output.  Usage: isa_by_line.py file.s kernel-substring [min_count]"""
import collections, re, sys
path, want = sys.argv[1], sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
files, cur, inside = {}, None, False
cnt = collections.defaultdict(lambda: collections.Counter())
for ln in open(path, errors="replace"):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)
    if m:
        files[int(m.group(1))] = m.group(3)
        continue
    if re.match(r"^_Z\w+:", ln):
        inside = want in ln
        continue
    if not inside:
        continue
    if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
        inside = False
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
        continue
    op = s.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    cnt[cur][kind] += 1
tot = collections.Counter()
for k, c in cnt.items():
    tot.update(c)
print("total", dict(tot))
for (f, l), c in sorted(cnt.items(), key=lambda kv: (kv[0][0] or "", kv[0][1])):
    if sum(c.values()) >= minc:
        print(f"{f}:{l:<5d} valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:3d} vmem {c['vmem']:3d}")
