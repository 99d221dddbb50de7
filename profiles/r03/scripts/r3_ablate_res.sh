cd /root/repo
export NMFK_STREAMS=1
for range in "4 4 256" "16 16 256" "2 16 32"; do
  echo "== k/restarts $range, 300 iterations"
  timeout -k 10 100 python scripts/microbench.py 300 $range
  for a in 32 64 128 192 4; do
    NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_abl$a.so timeout -k 10 100 python scripts/microbench.py 300 $range | sed "s#^.*libnmfk_hip_abl#abl#"
  done
done
