for a in 6 7 8 9; do for dbg in 3 0; do
  export TABMAT_AMD_LIB=$PWD/tabmat_amd/_abl/lib_abl$a.so
  echo "abl=$a (6 no fmac+no slab reads, 7 = 6 + no d read, 8 = 6 + fixed mask, 9 = all) dbg=$dbg"; TM_LG_DBG=$dbg python scripts/dev/time_k3_lg.py 10000000 f64 2>&1 | grep "lg unc=2"; done; done
