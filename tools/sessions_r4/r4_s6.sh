#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py "tests/test_gpu_e2e.py::test_cli_with_a_multi_million_line_barcode_list" "tests/test_gpu_e2e.py::test_cli_runs_again_when_the_umis_are_longer_than_the_64_bit_key_holds" "tests/test_gpu_e2e.py::test_cli_outputs_equal_oracle_bytes" -x -q > gpurun_out/r4/s6_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s6_tests.txt && rc=99
tail -40 gpurun_out/r4/s6_tests.txt
exit $rc
