mkdir -p profiles/_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DMPMPC_PHASE_CLOCK -Iinclude -o profiles/_ab/P.so multi-purpose-mpc_amd/csrc/mpmpc_hip.hip || exit 1
for tol in 1e-8 1e-7 2e-7; do
  for b in 1024 4096; do
    echo "== tol $tol B $b"; MPMPC_PHASES_SET="ipm_tol=$tol" python profiles/phases.py 2 $b | grep -E "interior point|active set|kernel body|ipm iterations|as rounds|as solves|percentiles|wave  "
  done
done
