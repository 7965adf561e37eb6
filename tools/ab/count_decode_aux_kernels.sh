cd /tmp && export TMPDIR=/tmp
for n in 32 96; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/dc$n -o dc -- python /root/repo/tools/profile_decode.py $n > /dev/null 2>&1
python - $n <<'PY'
import csv,glob,sys
f=glob.glob(f"/root/repo/gpurun_out/dc{sys.argv[1]}/*kernel_stats.csv")[0]
rows={r['Name'][:60]:int(r['Calls']) for r in csv.DictReader(open(f))}
print(sys.argv[1], {k:v for k,v in rows.items() if 'copyBuffer' in k or 'elementwise' in k or 'greedy' in k or 'fill' in k.lower()})
PY
rm -rf /root/repo/gpurun_out/dc$n
done
