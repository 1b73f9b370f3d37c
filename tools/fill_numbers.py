"""Fill the @@...@@ placeholders of DESIGN.md / README.md from the artefacts under profiles/ (round-end housekeeping)."""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.load(open(os.path.join(root, "profiles", "r04_bench.json")))
b3 = json.load(open(os.path.join(root, "profiles", "r04_bench_bands3.json")))
r = b["roofline"]; c = b["compactness_0.25"]; c4 = b["c4_whole_on_one_gpu"]; b9 = b["bands_9"]
tl = open(os.path.join(root, "profiles", "r04_step_timeline.txt")).readline()
vals = {"VALUE": f"{b['value']:.0f}", "MS": f"{b['ms_per_step']:.1f}", "SW": f"{r['avg_launch_ms']:.3f}", "FR": f"{r['frac']:.2f}",
        "RANGE": sys.argv[1] if len(sys.argv) > 1 else "5990–6050", "TLWALL": re.search(r"step wall ([0-9.]+)", tl).group(1),
        "C025": f"{c['value']:.0f}", "C025MS": f"{c['ms_per_step']:.1f}", "C025SW": f"{c['sweep_avg_launch_ms']:.3f}", "C025FR": f"{c['sweep_roofline_frac']:.2f}",
        "FP": f"{b['with_exit_on_fixed_point']['value']:.0f}", "C4": f"{c4['value']:.0f}", "C4MS": f"{c4['ms_per_step']:.1f}", "C4FR": f"{c4['sweep_roofline_frac']:.3f}",
        "B9": f"{b9['value']:.0f}", "B9MS": f"{b9['ms_per_step']:.1f}", "B9SW": f"{b9['sweep_avg_launch_ms']:.3f}", "B9FR": f"{b9['sweep_roofline_frac']:.2f}",
        "B3": f"{b3['value']:.0f}", "B3MS": f"{b3['ms_per_step']:.0f}"}
for f in ("DESIGN.md", "README.md"):
    p = os.path.join(root, f)
    s = open(p).read()
    for k, v in vals.items():
        s = s.replace("@@" + k + "@@", v)
    left = re.findall(r"@@[A-Z0-9]+@@", s)
    open(p, "w").write(s)
    print(f, "unfilled:", left)
