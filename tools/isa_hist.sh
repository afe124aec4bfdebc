#!/bin/bash
# instruction histogram of the kernels whose mangled name contains $2 in the -save-temps assembly $1
f=$1; pat=$2
for sym in $(grep -o "^_Z[A-Za-z0-9_]*${pat}[A-Za-z0-9_]*:" $f | tr -d ':' | sort -u); do
  echo "== $sym"
  awk -v s="$sym" '$0 ~ "^"s":"{on=1} on&&/s_endpgm/{print; exit} on' $f > /tmp/_k.s
  echo "instructions: $(grep -c "^\s*[vsdgb][a-z]*_" /tmp/_k.s)"
  grep -o "^\s*[a-z_0-9]*" /tmp/_k.s | sort | uniq -c | sort -rn | head -${3:-40} | awk '{printf "%s %s; ", $1, $2}'; echo
done
