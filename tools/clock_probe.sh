# Samples the GPU clocks / power while the planes GEMM micro-benchmark runs back to back (is the K loop power-bound?).
(for i in 1 2 3 4 5 6; do ./tools/micro/x3_gemm_bench_v0 8192 > /dev/null 2>&1; done) &
BG=$!
sleep 4
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|power|fclk" | head -6; echo ---; sleep 1; done
wait $BG
echo idle:
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|power" | head -4
