# build variant of dz_dense.hip only
out=$1; shift
R=/root/repo
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I $R/include "$@" -c $R/dqn_zoo_amd/csrc/dz_dense.hip -o $tmp/dz_dense.o 2>$tmp/err || { echo "FAILED $out"; grep -m3 error $tmp/err; rm -rf $tmp; exit 1; }
objs=$(ls $R/dqn_zoo_amd/csrc/_obj/*.o | grep -v dz_dense.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $out $tmp/dz_dense.o $objs && echo built $out
rm -rf $tmp
