"""Times tools/prof_target.py under each tuning build in gpurun_out/libs (development aid)."""
import glob
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["bistro", "20"]
for lib in sorted(glob.glob(os.path.join(root, "tuning_libs", "*.so"))):
    env = dict(os.environ, TRX_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "prof_target.py")] + args, env=env,
                         capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if "Mrays" in l]
    print(os.path.basename(lib), line[-1] if line else out.stderr[-300:], flush=True)
