run() { v=$1; shift; env "$@" FASTF_LIB_OVERRIDE=$v python bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | python tools/kline.py "$*"; }
for i in 8 9 10; do run fastf_amd/lib/libfastf_amd.so FASTF_SORT_IPT=$i; done
for i in 5 6 7 8; do run build/ipt8/libfastf_amd.so FASTF_SORT_IPT=$i V=ipt8; done
