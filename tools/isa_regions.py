"""dev: static instruction counts of the culled 1-NN kernel per marked region (NN_MARK in nn_compact.hpp).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGLOC_NN_MARKS -Iinclude -S --cuda-device-only \
        gloc3d_amd/csrc/reg.hip -o /tmp/regm.s && python tools/isa_regions.py /tmp/regm.s [kernel-name-prefix]
Counts are per region in PROGRAM order (the compiler moves blocks: read them next to the .s), static, not weighted by trip counts."""
import collections, re, sys

lines = open(sys.argv[1]).read().split("\n")
prefix = sys.argv[2] if len(sys.argv) > 2 else "_ZN4gloc3reg17nn_compact_kernelILi2ELb0ELb0E"
start = next(i for i, l in enumerate(lines) if l.startswith(prefix))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
cur, order, cnt = "entry", ["entry"], collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    t = l.strip()
    m = re.search(r"NN_MARK (\w+)", t)
    if m:
        cur = m.group(1)
        if cur not in order:
            order.append(cur)
        continue
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_")
            else "VMEM" if op.startswith(("global_", "flat_", "buffer_")) else "other")
    cnt[cur][kind] += 1
    if op.startswith("v_pk"):
        cnt[cur]["packed"] += 1
    if "_f64" in op:
        cnt[cur]["f64"] += 1
for n in order:
    print(f"{n:18s}", dict(cnt[n]))
