R=$PWD
mkdir -p gpurun_out/c2
python tools/c2_trace.py > gpurun_out/c2/wall.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c2/tr -o c2 -- python3 $R/tools/c2_trace.py > /dev/null 2>&1
cd $R
DB=$(find gpurun_out/c2/tr -name "*.db" | head -1)
python tools/c2_timeline.py $DB > gpurun_out/c2/timeline.txt 2>&1
rm -rf gpurun_out/c2/tr
