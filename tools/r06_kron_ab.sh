for i in 1 2 3; do
PSGD_HIP_LIB=$PWD/build_ab/libpsgd_hip_old.so python tools/r06_kron_ab.py old
python tools/r06_kron_ab.py new
done
