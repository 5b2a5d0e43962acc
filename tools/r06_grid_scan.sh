for cfg in "12500032 20" "50000000 10" "20000000 32" "100000000 8" "4000000 20" "30000000 16"; do
set -- $cfg
echo "== N $1 r $2 (packed)"
GS_N=$1 GS_R=$2 GS_PLACE=0 GS_GRIDS=256,512 python tools/r06_grid_scan.py 2>&1 | grep -v "amdgpu.ids\|layout"
done
