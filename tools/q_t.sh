cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d /tmp/p3 -o p -- python /root/repo/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-tape > /tmp/c3.json 2> /tmp/c3.err
python3 - <<'PY'
import csv, glob, json
f = glob.glob('/tmp/p3/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); w = rows[int(n*0.8)+30:int(n*0.8)+62]
t0 = int(w[0]["Start_Timestamp"])
for r in w:
    print(r["Queue_Id"], r["Kernel_Name"][:34].ljust(34), round((int(r["Start_Timestamp"])-t0)/1e3,1), round((int(r["End_Timestamp"])-t0)/1e3,1))
PY
