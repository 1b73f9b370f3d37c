"""profiles/r04_traffic.json from two rocprofv3 --pmc passes of tools/step_trace.py (the bench workload):
     rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/step_trace.py
     rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/step_trace.py
     python3 tools/traffic_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write <commit> > profiles/r04_traffic.json
Units: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md,
HBM section; calibrated on features_kernel, which reads and writes exactly 32 B/pixel) -- doubled here."""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha256


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
commit = sys.argv[3] if len(sys.argv) > 3 else None
S, T, B, C = 16384, 2048, 64, 8
nt = S // T
win = lambda i: min(S, i * T + T + B) - max(0, i * T - B)      # noqa: E731
black_px = sum(T * T for tj in range(nt) for ti in range(nt) if (ti + tj) % 2 == 0)
white_px = sum(win(tj) * win(ti) for tj in range(nt) for ti in range(nt) if (ti + tj) % 2 == 1)
px_per_launch = (black_px + white_px) / (1 + nt)               # one black batch + one batch per white tile row
names = {"slic_assign_colour": "slic_assign_kernel<8, true, false, false, false", "slic_prepass": "slic_prepass_kernel<8, true, false>",
         "features": "features_planes_kernel<8,", "band_minmax": "band_minmax_kernel<4>", "zonal": "zonal_kernel<8>"}
out = {"commit": commit, "kernel_source_sha256": kernel_source_sha256(),
       "workload": {"size": S, "tile": T, "buffer": B, "bands": C, "compactness": 10.0},
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/step_trace.py; bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024, mean over dispatches",
       "kernels": {}}
for key, pat in names.items():
    fk = [k for k in fetch if pat in k]
    wk = [k for k in write if pat in k]
    if not fk or not wk:
        continue
    f, nf = fetch[fk[0]]
    w, _ = write[wk[0]]
    e = {"dispatches": nf, "fetch_size_kib": f, "write_size_kib": w, "read_bytes": 2 * f * 1024, "write_bytes": w * 1024,
         "bytes_per_launch": 2 * f * 1024 + w * 1024}
    if key in ("slic_assign_colour", "slic_prepass"):
        e["pixels_per_launch"] = px_per_launch
        e["bytes_per_pixel"] = e["bytes_per_launch"] / px_per_launch
    out["kernels"][key] = e
print(json.dumps(out, indent=1))
