# Experiment build of the library with extra -D flags:  bash tools/build_exp.sh NAME -DFLAG ...  -> build_exp/lib_NAME.so
set -e
cd "$(dirname "$0")/../pylatticedso_amd/csrc"
NAME=$1; shift
mkdir -p ../../build_exp
make -s pl_hostgen.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -c pl_api.hip -o ../../build_exp/pl_api_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC ../../build_exp/pl_api_$NAME.o pl_hostgen.o -o ../../build_exp/lib_$NAME.so -shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -pthread
rm -f ../../build_exp/pl_api_$NAME.o
echo built build_exp/lib_$NAME.so
