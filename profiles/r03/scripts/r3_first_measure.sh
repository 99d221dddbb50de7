set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -q -m gpu -x > gpurun_out/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t2.log
tail -5 gpurun_out/t2.log
{
echo "== default (round-3 variants, all ranks on MFMA)"; timeout -k 10 120 python scripts/microbench.py 400 2 16 32
echo "== serial streams"; NMFK_STREAMS=1 timeout -k 10 120 python scripts/microbench.py 400 2 16 32
echo "== round-2 schedule (NMFK_HYB_SMALL=0)"; NMFK_HYB_SMALL=0 timeout -k 10 120 python scripts/microbench.py 400 2 16 32
echo "== NS3 on"; NMFK_HYB_NS3=1 timeout -k 10 120 python scripts/microbench.py 400 2 16 32
echo "== per variant alone: k 2..4"; timeout -k 10 120 python scripts/microbench.py 400 2 4 32
echo "== k 5..8"; timeout -k 10 120 python scripts/microbench.py 400 5 8 32
echo "== k 9..12"; timeout -k 10 120 python scripts/microbench.py 400 9 12 32
echo "== k 9..12 NS3"; NMFK_HYB_NS3=1 timeout -k 10 120 python scripts/microbench.py 400 9 12 32
echo "== k 13..16"; timeout -k 10 120 python scripts/microbench.py 400 13 16 32
} > gpurun_out/mb1.log 2>&1
cat gpurun_out/mb1.log
