#!/bin/bash
# A/B of two BUILDS of the library on one box (run there): tools/build_ab.sh "<hipcc flags of B>" <command...>
# runs the command on the build without the flags and on the build with them, alternating twice.
FLAGS="$1"; shift
for rep in 1 2; do
  python video-query-algorithms_amd/build.py --force > /dev/null 2>&1 && echo "== A (product build)" && "$@"
  VQ_EXTRA_HIPCC_FLAGS="$FLAGS" python video-query-algorithms_amd/build.py --force > /dev/null 2>&1 && echo "== B ($FLAGS)" && "$@"
done
python video-query-algorithms_amd/build.py --force > /dev/null 2>&1
