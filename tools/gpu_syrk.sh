#!/bin/bash
# weighted syrk at the headline width: workgroup stamps of its main loop beside the sibling products (diagnostic build
# abtest/libstamps.so: tools/build_variant.sh stamps -DGEMM_STAMPS), and its HBM-side traffic (two PMC passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/syrk; mkdir -p $O gpurun_out/chol
MOBOCMF_HIP_LIB=$PWD/abtest/libstamps.so timeout -k 10 200 python tools/gemm_stamps.py 512 16384 2>&1 | grep -v amdgpu.ids > $O/stamps_16384.txt || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_syrk.py 512 16384 > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/pmc_syrk.py 512 16384 > /dev/null 2>&1 &&
python tools/pmc_syrk_summarize.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/pmc_syrk.json 512 16384 > /dev/null
rm -rf $O/pmc_fetch $O/pmc_write
cat $O/pmc_syrk.json | head -40
