mkdir -p gpurun_out/r02
for cb in 131072 196608 262144 327680 393216 524288; do
  for cfg in "64 400000 6" "64 400000 1" "8 400000 1" "256 20000 1"; do
    echo -n "chunk $cb cfg $cfg: " >> gpurun_out/r02/sweep.log
    VKIMG_GZ_CHUNK_BYTES=$cb timeout -k 10 120 python tools/inflate_time.py $cfg 2>&1 | grep -o "GPU inflate [0-9.]* ms = [0-9.]* GB/s" >> gpurun_out/r02/sweep.log || exit 1
  done
done
