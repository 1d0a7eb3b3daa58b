#!/bin/bash
# A = shipped library, B = a variant in profiles/_ab/B.so: the tolerance sweep on both, same box
D=multi-purpose-mpc_amd/csrc
cp $D/libmpmpc.so /tmp/keep.so
for v in A B; do
  cp profiles/_ab/$v.so $D/libmpmpc.so
  echo "=== library $v"
  bash profiles/sweep_tol.sh ${KEY:-ipm_tol} "$@"
done
cp /tmp/keep.so $D/libmpmpc.so
