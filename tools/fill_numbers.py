"""Fill the @@...@@ placeholders of DESIGN.md / README.md from the artefacts under profiles/ (round-end housekeeping).
   python tools/fill_numbers.py [round prefix, default r04]"""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda f: os.path.join(root, "profiles", f"{R}_{f}")   # noqa: E731
b = json.load(open(P("bench.json")))
b3 = json.load(open(P("bench_bands3.json"))) if os.path.exists(P("bench_bands3.json")) else None
c4f = json.load(open(P("bench_c4.json"))) if os.path.exists(P("bench_c4.json")) else None
r = b["roofline"]; s = b["stage_ms_per_step"]
c = b.get("compactness_0.25") or {}; c4 = b.get("c4_whole_on_one_gpu") or {}; b9 = b.get("bands_9") or {}
fp = b.get("with_exit_on_fixed_point") or {}; qs = b.get("quickshift") or {}; nr = b.get("next_rows") or {}; cpu = b.get("cpu_baseline") or {}
tl = open(P("step_timeline.txt")).readline() if os.path.exists(P("step_timeline.txt")) else ""
tr = json.load(open(P("traffic.json"))) if os.path.exists(P("traffic.json")) else {}
so = open(P("shard_overhead.txt")).read() if os.path.exists(P("shard_overhead.txt")) else ""
g = lambda d, k, f="{:.0f}": (f.format(d[k]) if d and d.get(k) is not None else "n/a")   # noqa: E731
m = lambda pat, txt: (re.search(pat, txt).group(1) if re.search(pat, txt) else "n/a")   # noqa: E731
c4ms = c4.get("ms_per_step")
vals = {"VALUE": f"{b['value']:.0f}", "MS": f"{b['ms_per_step']:.1f}", "SW": f"{r['avg_launch_ms']:.4f}", "FR": f"{r['frac']:.3f}",
        "PRE": f"{s['prepass_ms']:.1f}", "ASSIGN": f"{s['assign_ms']:.1f}", "CC": f"{s['connectivity_ms']:.2f}", "FEAT": f"{s['features_ms']:.1f}",
        "ZONAL": f"{s['zonal_ms']:.2f}", "TLWALL": m(r"step wall ([0-9.]+)", tl), "LAUNCHES": m(r"launches ([0-9]+)", tl), "IDLE": m(r"idle ([0-9.]+)", tl),
        "TRAFFIC": (f"{tr['kernels']['slic_assign_colour']['bytes_per_pixel']:.2f}" if tr.get("kernels", {}).get("slic_assign_colour") else "n/a"),
        "C025": g(c, "value"), "C025MS": g(c, "ms_per_step", "{:.1f}"), "C025SW": g(c, "sweep_avg_launch_ms", "{:.4f}"), "C025FR": g(c, "sweep_roofline_frac", "{:.3f}"),
        "FP": g(fp, "value"), "C4": g(c4, "value"), "C4MS": g(c4, "ms_per_step", "{:.1f}"), "C4FR": g(c4, "sweep_roofline_frac", "{:.3f}"),
        "C4IDEAL": (f"{c4ms / 8:.1f}" if c4ms else "n/a"), "C4BUDGET": (f"{c4ms / 6:.1f}" if c4ms else "n/a"),
        "B9": g(b9, "value"), "B9MS": g(b9, "ms_per_step", "{:.1f}"), "B9SW": g(b9, "sweep_avg_launch_ms", "{:.4f}"), "B9FR": g(b9, "sweep_roofline_frac", "{:.3f}"),
        "B3": g(b3, "value"), "B3MS": g(b3, "ms_per_step", "{:.1f}"), "QS": g(qs, "value"),
        "B3X": (f"{b3['ms_per_step'] / b['ms_per_step']:.2f}" if b3 else "n/a"),
        "MOM": g(nr, "moments_ms", "{:.2f}"), "MOMGB": g(nr, "moments_GBps"), "TEX": g(nr, "texture_one_band_ms", "{:.1f}"),
        "CPU1": g(cpu, "value", "{:.2f}"), "CPU16": g(cpu.get("all_cores"), "value", "{:.1f}"), "CPULIB": g(cpu.get("library"), "value", "{:.2f}"),
        "IMPORT": (so.strip().splitlines()[0] if so.strip() else "n/a"),
        "IMPORTMS": m(r"repeat import ([0-9.]+) ms", so)}
for f in ("DESIGN.md", "README.md"):
    p = os.path.join(root, f)
    t = open(p).read()
    for k, v in vals.items():
        t = t.replace("@@" + k + "@@", v)
    left = re.findall(r"@@[A-Z0-9]+@@", t)
    open(p, "w").write(t)
    print(f, "unfilled:", left)
