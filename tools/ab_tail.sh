#!/bin/bash
for lib in "$@"; do echo "== $lib"; RT_LIB=$PWD/$lib timeout -k 10 300 python -u tools/probe_tail.py 2>&1 | grep -v amdgpu.ids; done
