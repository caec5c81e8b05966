# Build variants of libibhip.so with extra compiler flags for same-box A/B runs (IBHIP_LIB selects one):
#   bash scripts/ab_build.sh name1 "-DX=1" name2 "-DX=2" ...   -> immersedboundary.jl_amd/lib_<name>.so
set -e
D=immersedboundary.jl_amd/csrc
BASE=$(grep '^CXXFLAGS :=' $D/Makefile | sed 's/^CXXFLAGS := //; s/\$(ARCH)/gfx950/')
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  make -C $D clean >/dev/null
  make -C $D -j8 CXXFLAGS="$BASE $f" >/dev/null 2>&1
  cp immersedboundary.jl_amd/libibhip.so immersedboundary.jl_amd/lib_$n.so
done
make -C $D clean >/dev/null; make -C $D -j8 >/dev/null 2>&1
