# build_variants/lib_NAME.so: the library with extra hipcc flags, for tools/run_variants.sh
#   bash tools/build_variant.sh NAME [flags...]     e.g.  bash tools/build_variant.sh w10 -DSG_W=10
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build_variants
make -s -C $R/zsc_amd/csrc $R/zsc_amd/csrc/zsc_api.o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -mllvm --disable-machine-licm -I$R/include -I$R/zsc_amd/csrc "$@" \
    -c $R/zsc_amd/csrc/zsc_hip_runtime.hip -o $R/build_variants/rt_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/lib_$name.so $R/build_variants/rt_$name.o $R/zsc_amd/csrc/zsc_api.o
rm -f $R/build_variants/rt_$name.o
echo built lib_$name.so
