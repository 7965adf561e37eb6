cd /tmp && export TMPDIR=/tmp
for B in 4 8; do
BATCHES=$B timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/bd$B -o bd -- python /root/repo/tools/batched_decode.py > /root/repo/gpurun_out/bd$B.log 2>&1
find /root/repo/gpurun_out/bd$B -name "*kernel_trace.csv" -delete; find /root/repo/gpurun_out/bd$B -name "*agent_info.csv" -delete
done
