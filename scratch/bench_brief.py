"""Run bench.py with the given arguments and print the few numbers an A/B needs."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-shapes", *sys.argv[1:]], capture_output=True, text=True)
lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not lines:
    print("FAILED", out.stderr[-1500:])
    sys.exit(1)
d = json.loads(lines[-1])
r = d["roofline"] or {}
print(f"{os.environ.get('FAIRREC_HIP_LIB', 'product')[-14:]:>14s} {' '.join(sys.argv[1:]):50s} us/step {d['ms_per_step'] * 1e3:7.2f}  kernels {r.get('kernel_us')}  modes {d['config'].get('launch_modes_timed')}")
