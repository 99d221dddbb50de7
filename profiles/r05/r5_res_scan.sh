python scripts/r5_geo_scan.py 32 "-1,-1" "-1,-1,-1;-1,-1,2;-1,-1,3;-1,-1,4;-1,-1,6" 1
python scripts/r5_geo_scan.py 16 "-1,-1" "-1,-1,-1;-1,-1,2;-1,-1,3;-1,-1,4" 1
