mkdir -p gpurun_out/r05a
export MTH_PTILE_STATIC=1
{
for w in 1 2 3 4 5 6 7 8 10; do echo "static wgs=$w $(MTH_PTILE_WGS_PER_CU=$w python tools/time_tile.py 100 2>&1 | tail -1)"; done
echo "nodeep wgs=8 $(METHEOR_HIP_LIB=$PWD/abx/libnodeep.so python tools/time_tile.py 100 2>&1 | tail -1)"
echo "nodeep wgs=4 $(MTH_PTILE_WGS_PER_CU=4 METHEOR_HIP_LIB=$PWD/abx/libnodeep.so python tools/time_tile.py 100 2>&1 | tail -1)"
echo "pdr only $(ONLY=pdr python tools/time_tile.py 100 2>&1 | tail -1)"
echo "lpmd only $(ONLY=lpmd python tools/time_tile.py 100 2>&1 | tail -1)"
echo "old pdr only $(MTH_TILE_PERSIST=0 ONLY=pdr python tools/time_tile.py 100 2>&1 | tail -1)"
echo "old lpmd only $(MTH_TILE_PERSIST=0 ONLY=lpmd python tools/time_tile.py 100 2>&1 | tail -1)"
} > gpurun_out/r05a/time2.log 2>&1
cat gpurun_out/r05a/time2.log
