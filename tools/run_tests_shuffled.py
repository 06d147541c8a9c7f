"""Run the GPU tests in reversed and in shuffled orders inside one process each (order dependence = shared state,
uninitialised memory or lifetime bugs that a fixed order hides): tools/run_tests_shuffled.py [n_shuffles]"""
import random
import subprocess
import sys

ids = [l.strip() for l in subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--co", "-q"],
                                         capture_output=True, text=True).stdout.splitlines() if "::" in l]
print(len(ids), "tests")
orders = [("reversed", list(reversed(ids)))]
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    o = ids[:]
    random.Random(s).shuffle(o)
    orders.append((f"shuffle {s}", o))
bad = 0
for name, o in orders:
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + o, capture_output=True, text=True)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]
    print(f"{name}: {tail}")
    if r.returncode != 0:
        bad += 1
        print("\n".join(r.stdout.splitlines()[-25:]))
sys.exit(1 if bad else 0)
