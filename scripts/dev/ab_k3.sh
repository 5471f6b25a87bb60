# A/B of two builds of the K3 entry kernel on ONE box, interleaved: default vs tabmat_amd/_abl/libtabmat_$1.so
for r in 1 2 3; do
  python scripts/time_k3.py 2>&1 | grep K3 | sed "s/^/default  /"
  TABMAT_AMD_LIB=tabmat_amd/_abl/libtabmat_$1.so python scripts/time_k3.py 2>&1 | grep K3 | sed "s/^/$1  /"
done
