# blocking-sync flag through torch's own HIP runtime: does a host wait sleep?  (CPU seconds of a process that waits ~3 s for the device)
python - <<'PY'
import os, sys, time, resource
sys.path.insert(0, os.getcwd())
import bench, torch
flag = os.environ.get("FLAG") == "1"
if flag: print("hipSetDeviceFlags ok:", bench.blocking_sync(0))
x = torch.zeros(1, device="cuda")
a = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
c0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.time()
for _ in range(60): b = a @ a
v = b[0, 0].item()
c1 = resource.getrusage(resource.RUSAGE_SELF); t1 = time.time()
print("flag", flag, "wall %.2f s, cpu user+sys %.2f s" % (t1 - t0, (c1.ru_utime + c1.ru_stime) - (c0.ru_utime + c0.ru_stime)))
PY
