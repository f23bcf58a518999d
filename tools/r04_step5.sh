# round 4, step 5: where a replay launch spends its time.  gpurun --timeout 900 -- 'bash tools/r04_step5.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/prof_replay
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_replay -o r --output-format csv -- python3 tools/replay_timing.py --batch ${1:-16} --reps 1 > gpurun_out/r04_replay_timing.txt 2>&1 || { tail -5 gpurun_out/r04_replay_timing.txt; exit 1; }
tail -2 gpurun_out/r04_replay_timing.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_replay/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    if "k_replay" in n or "expand_fast" in n:
        out.append((n[-30:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size") or r.get("Grid_Size_X")))
for o in out[-2 * 12:]:
    print(o)
PY
rm -f gpurun_out/prof_replay/*/*kernel_trace.csv
