cd /root/repo
run() { echo "== $1"; shift; env "$@" timeout -k 10 150 python scripts/microbench.py 300 $KR; }
KR="2 16 32"; run "default" A=1
KR="2 16 32"; run "serial" NMFK_STREAMS=1
KR="2 4 32"; run "k2-4 x32 serial" NMFK_STREAMS=1
KR="2 4 96"; run "k2-4 x96 serial" NMFK_STREAMS=1
KR="4 4 256"; run "k4 x256 serial" NMFK_STREAMS=1
KR="8 8 256"; run "k8 x256 serial" NMFK_STREAMS=1
KR="16 16 256"; run "k16 x256 serial" NMFK_STREAMS=1
KR="12 12 256"; run "k12 x256 serial NS3" NMFK_STREAMS=1 NMFK_HYB_NS3=1
KR="12 12 256"; run "k12 x256 serial 16-signal" NMFK_STREAMS=1
