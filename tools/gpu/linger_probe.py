"""Which process keeps an inherited pipe open after `torchrun bench.py` exits?"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
env = dict(os.environ, UCSA_BENCH_BACKEND="gloo")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
       "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "bench.py"),
       "--gpus", "2", "--mode", "train", "--steps", "2", "--warmup", "1", "--pretrain-steps", "30"]
t0 = time.time()
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT)
rc = p.wait()
print("torchrun exited rc", rc, "after", round(time.time() - t0, 1), "s", flush=True)
print(subprocess.run("ps -eo pid,ppid,etimes,stat,cmd --forest | grep -v 'ps -eo' | tail -30", shell=True,
                     capture_output=True, text=True).stdout, flush=True)
# who holds our pipe?
my = os.readlink(f"/proc/self/fd/{p.stdout.fileno()}")
print("pipe", my)
for pid in os.listdir("/proc"):
    if pid.isdigit() and int(pid) != os.getpid():
        try:
            for fd in os.listdir(f"/proc/{pid}/fd"):
                if os.readlink(f"/proc/{pid}/fd/{fd}") == my.replace("pipe", "pipe"):
                    print("held by", pid, open(f"/proc/{pid}/cmdline").read().replace("\0", " ")[:200])
                    break
        except OSError:
            pass
