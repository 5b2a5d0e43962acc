./tools/micro/p3_data_probe > gpurun_out/p3data.txt 2>&1 &
BG=$!
for i in $(seq 1 12); do sleep 1; echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E 'sclk|Power \(W\)' | sed 's/.*: //' | tr '\n' ' ')"; done
wait $BG
cat gpurun_out/p3data.txt
