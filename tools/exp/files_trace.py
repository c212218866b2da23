"""Device-side timeline of ONE LBAudioDetectiveProcessAudioURLs call of copies x sixty fixtures, from a
`rocprofv3 --kernel-trace --memory-copy-trace` run of tools/exp/files_ctime.py (csv directory as argument): kernels and copies in
start order with their gaps, for the last call in the trace."""
import csv, glob, os, sys
root = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?")) + " " + r.get("Bytes", r.get("Size", "?"))))
ev.sort()
t_end = ev[-1][1]
win = [e for e in ev if e[0] > t_end - int(float(os.environ.get("WINDOW_MS", "70")) * 1e6)]
t0 = win[0][0]
last_k_end = t0
busy = 0
for s, e, n in win:
    if n.startswith("K "):
        gap = s - last_k_end
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {n}")
        last_k_end = max(last_k_end, e)
        busy += e - s
    elif (e - s) > 200e3 or os.environ.get("ALL_COPIES"):
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us               {n}")
print("kernel time in the window:", busy / 1e6, "ms of", (win[-1][1] - t0) / 1e6)
