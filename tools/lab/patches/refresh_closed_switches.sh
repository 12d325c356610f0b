#!/bin/bash
# Re-bases tools/lab/patches/closed_switches.patch on the current csrc/ after the product source has moved: applies the old patch to a scratch copy
# (patch's fuzz carries most edits along), stops on rejects -- fix those by hand in the scratch copy ($T/b) and run again with --finish -- and writes
# the new patch as the diff from the current tree to the switched copy.  tests/test_kernel_math_host.py::test_closed_switches_patch_applies_and_is_neutral
# says when this is needed.
set -e
root=$(cd "$(dirname "$0")/../../.." && pwd)
T=${TMPDIR:-/tmp}/closed_switches_refresh
if [ "$1" != "--finish" ]; then
  rm -rf $T; mkdir -p $T/a/quadruped_drake_amd $T/b/quadruped_drake_amd
  cp -r $root/quadruped_drake_amd/csrc $T/a/quadruped_drake_amd/; cp -r $root/quadruped_drake_amd/csrc $T/b/quadruped_drake_amd/
  (cd $T/b && patch -p1 --no-backup-if-mismatch -i $root/tools/lab/patches/closed_switches.patch) || { echo "rejects under $T/b: merge them by hand, then: $0 --finish"; exit 1; }
fi
find $T/b -name "*.rej" -o -name "*.orig" | xargs rm -f
(cd $T && diff -u -r a b | grep -v "^diff -u\|^Only in" | sed -E 's/^(---|\+\+\+) ([^\t]*)\t.*/\1 \2/' > $root/tools/lab/patches/closed_switches.patch) || true
echo "closed_switches.patch: $(grep -c '^@@' $root/tools/lab/patches/closed_switches.patch) hunks"
