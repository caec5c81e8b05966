# Diagnostic builds of the tuned block path (never shipped): what the sweep's time is made of.
# usage (on the GPU box): bash scripts/ablate_passB.sh base noxcd nomath nohalo neither
set -e
B="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950"
run() { make -C immersedboundary.jl_amd/csrc clean >/dev/null; make -C immersedboundary.jl_amd/csrc -j8 CXXFLAGS="$B $2" >/dev/null 2>&1; python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'sweep_us', round(d['ms_per_step']*1e3,2), 'passB_us', d['roofline']['kernel_us'], 'passA_us', d['roofline']['passA_us'])"; }
for v in "$@"; do
  case $v in
    base) run base "";;
    noxcd) run noxcd "-DIBH_NO_XCD_REMAP";;
    nomath) run nomath "-DIBH_ABLATE_NOMATH";;
    nohalo) run nohalo "-DIBH_ABLATE_NOHALO";;
    neither) run neither "-DIBH_ABLATE_NOMATH -DIBH_ABLATE_NOHALO";;
  esac
done
make -C immersedboundary.jl_amd/csrc clean >/dev/null; make -C immersedboundary.jl_amd/csrc -j8 >/dev/null 2>&1
