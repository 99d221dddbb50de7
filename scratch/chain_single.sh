for k in 16 4; do for c in 2 1; do echo "k=$k chain=$c: $(NMFK_CHAIN=$c timeout 100 python scripts/microbench.py 200 $k $k 4 | cut -c26-60)"; done; done
for c in 2 1; do echo "k=9..16 chain=$c: $(NMFK_CHAIN=$c timeout 100 python scripts/microbench.py 200 9 16 4 | cut -c26-60)"; done
