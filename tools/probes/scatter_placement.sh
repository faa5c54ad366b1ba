# gpurun -- bash tools/probes/scatter_placement.sh
export TMPDIR=/tmp
mkdir -p gpurun_out/placement
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/placement/trace -- python3 tools/probes/scatter_placement.py > gpurun_out/placement/addr.txt 2>&1
cat gpurun_out/placement/addr.txt | grep workspace
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/placement/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for name in ('scatter_kernel', 'match_kernel'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if name in r['Kernel_Name']]
    print(name, 'per dispatch, us (6 workspaces x 3 rounds):')
    for i in range(0, len(d), 6):
        print('   ', ' '.join(f'{x:7.1f}' for x in d[i:i + 6]))
PY
find gpurun_out/placement -name "*.csv" -delete; find gpurun_out/placement -name "*.db" -delete
