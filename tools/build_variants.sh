#!/bin/bash
# builds experiment libraries side by side: tools/build_variants.sh name1 "flags1" name2 "flags2" ...  -> build_ab/<name>.so
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  make -s -j3 LIB=build_ab/$n.so OBJDIR=build_ab/$n EXTRA="$f" product > build_ab/$n.build.log 2>&1 &
  # two builds at a time: 8 cores
  if (( $(jobs -r | wc -l) >= 3 )); then wait -n; fi
done
wait
ls -la build_ab/*.so
