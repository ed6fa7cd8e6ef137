L=$PWD/ubisoft-laforge-zeroeggs_amd/zeggs
for v in hip NOX NOMFMA NOBOTH; do echo "== $v"; ZEGGS_LIB=$L/libzeggs_$v.so python tools/stage_bench.py 2>&1 | grep "variant  0"; done
