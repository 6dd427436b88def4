"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<name>.json (HBM bytes per launch per kernel).
  python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
gfx950 corrections per MI355X_MICROARCH.md: FETCH_SIZE (KiB) x 1024 x 2, WRITE_SIZE (KiB) x 1024."""
import csv, glob, json, sys, collections, re
fd, wd, out = sys.argv[1:4]
def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:110]
            a = acc[n]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc
F = load(fd, "FETCH_SIZE"); W = load(wd, "WRITE_SIZE")
ker = {}
for n in sorted(set(F) | set(W)):
    if n.startswith("void at::") or "rocclr" in n or n.startswith("at::"): continue
    lf, vf = F.get(n, [0, 0.0]); lw, vw = W.get(n, [0, 0.0])
    ker[n] = {"launches": max(lf, lw), "fetch_bytes_per_launch": round(vf * 1024 * 2 / max(lf, 1)), "write_bytes_per_launch": round(vw * 1024 / max(lw, 1))}
note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline` "
        "(tools/pmc_traffic.sh). Units: bytes per launch, averaged over the launches of that kernel. gfx950 correction per MI355X_MICROARCH.md: "
        "FETCH_SIZE (KiB) x 1024 x 2 (wide coalesced streams are tallied at half); WRITE_SIZE (KiB) x 1024. Cross-check: adamw_kernel should "
        "read 16 B/param and write 18 B/param x 863.2 M params.")
json.dump({"_note": note, "kernels": ker}, open(out, "w"), indent=1)
for n in ker:
    if "adamw" in n or "gemm_kernel<256" in n: print(n[:90], ker[n])
