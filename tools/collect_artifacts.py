"""After `gpurun -- 'bash tools/final_artifacts.sh'`: profiles/<round>_traffic.json from the newest counter files, the bench lines and
the kernel statistics copied under profiles/.   python tools/collect_artifacts.py <commit> [round prefix, default r04]"""
import csv, glob, json, os, shutil, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[2] if len(sys.argv) > 2 else "r04"
go = os.path.join(root, "gpurun_out")
newest = lambda pat: sorted(glob.glob(os.path.join(go, pat)), key=os.path.getmtime)[-1]   # noqa: E731
tf, tw = tempfile.mkdtemp(), tempfile.mkdtemp()
shutil.copy(newest("fa_fetch/*/*_counter_collection.csv"), tf)
shutil.copy(newest("fa_write/*/*_counter_collection.csv"), tw)
out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "traffic_json.py"), tf, tw, sys.argv[1]])
open(os.path.join(root, "profiles", R + "_traffic.json"), "wb").write(out)
t = json.loads(out)
print(t["commit"], {k: round(v.get("bytes_per_pixel", 0), 2) for k, v in t["kernels"].items()})
for src, dst in (("fa_bench.json", R + "_bench.json"), ("fa_bench_rocprof.json", R + "_bench_under_rocprof.json"), ("fa_bench_c4.json", R + "_bench_c4.json")):
    b = json.loads(open(os.path.join(go, src)).read().strip().splitlines()[-1])
    r = b["roofline"]
    print(dst, b["value"], b["ms_per_step"], r["avg_launch_ms"], r["frac"], r["traffic"],
          {k: (b[k] or {}).get("value") for k in ("with_exit_on_fixed_point", "compactness_0.25", "quickshift", "cpu_baseline")})
    shutil.copy(os.path.join(go, src), os.path.join(root, "profiles", dst))
for src, dst in (("fa_bench_bands3.json", R + "_bench_bands3.json"), ("fa_step_timeline.txt", R + "_step_timeline.txt"), ("fa_step_timeline_c025.txt", R + "_step_timeline_c025.txt")):
    if os.path.exists(os.path.join(go, src)):
        shutil.copy(os.path.join(go, src), os.path.join(root, "profiles", dst))
ks = newest("fa_trace/*/*_kernel_stats.csv")
for row in list(csv.DictReader(open(ks)))[:4]:
    print(row["Name"][:70], row["Calls"], row["AverageNs"])
shutil.copy(ks, os.path.join(root, "profiles", R + "_kernel_stats.csv"))
