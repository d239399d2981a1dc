#!/bin/bash
# tools/patch_variant.sh <name> <patch> [hipcc flags...]: an A/B build of trace.hip with one of tools/patches/*.patch applied (the source tree is restored afterwards)
set -e
N=$1; P=$2; shift 2
git apply "$P"
trap 'git checkout embree_amd/csrc/trace.hip' EXIT
tools/trace_variant.sh "$N" "$@"
