# stream order of the large update (key 25) and dX planes on the side stream (key 29) under tile scales: bash tools/r05_order_ab.sh
for shape in "4096 4096" "2048 4096" "4096 2048" "3072 3072" "6144 6144" "2304 2048" "8192 2048"; do
  for keys in "25:0" "25:1" "25:1,29:0" "25:-1"; do
    KRON_KEYS=$keys python tools/r04_time_update.py $shape 2>&1 | tail -1
    KRON_KEYS=$keys python tools/r04_time_update.py $shape bf16 2>&1 | tail -1
  done
done
