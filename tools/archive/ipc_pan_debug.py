import sys, os, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")   # run from the repo root: python tools/archive/ipc_pan_debug.py [reps] [options json]; DBG=1 prints the first diverging trace line
import test_gpu_parity as t
import subprocess, json, uuid
dbg = os.environ.get("DBG", "0")
opts = sys.argv[2] if len(sys.argv) > 2 else json.dumps({"panel": 1})
nfail = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    key = uuid.uuid4().hex
    procs = []
    t0 = time.time()
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MACHIP_RDZV_KEY=key, HSA_ENABLE_IPC_MODE_LEGACY="0", IPC_TIMEOUT="5")
        if dbg != "0": env["MACHIP_DEBUG"] = dbg
        procs.append(subprocess.Popen([sys.executable, "-c", t.IPC_WORKER, "c4", "5", "-1", opts], env=env, cwd=t.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=200) for p in procs]
    rcs = [p.returncode for p in procs]
    print("rep", rep, "rc", rcs, "%.1fs" % (time.time() - t0), flush=True)
    if any(rcs):
        nfail += 1
        if dbg != "0":
            a = [l for l in outs[0][1].splitlines() if l.startswith("[machip]")]
            b = [l for l in outs[1][1].splitlines() if l.startswith("[machip]")]
            for i, (x, y) in enumerate(zip(a, b)):
                if x != y and "us per launch" not in x:
                    print("first difference at trace line", i)
                    for k in range(max(0, i - 4), min(len(a), i + 3)): print("  r0:", a[k][:200])
                    for k in range(max(0, i - 4), min(len(b), i + 3)): print("  r1:", b[k][:200])
                    break
print("failed reps:", nfail)
