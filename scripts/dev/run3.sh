for a in 3 5; do for dbg in 3 0; do
  export TABMAT_AMD_LIB=$PWD/tabmat_amd/_abl/lib_abl$a.so
  echo "abl=$a (3 no slab reads, 5 half the slab reads) dbg=$dbg"; TM_LG_DBG=$dbg python scripts/dev/time_k3_lg.py 10000000 f64 2>&1 | grep "lg unc=2"; done; done
