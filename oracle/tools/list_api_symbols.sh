#!/bin/bash
# Names of the functions the reference's public header declares -> tests/golden/mbelib_api_symbols.txt (data: a list of
# API names; authoring container only, reads the header where it lies).
REF=${REF:-/root/reference}
grep -oE "\bmbe_[A-Za-z0-9_]+\s*\(" "$REF/include/mbelib-neo/mbelib.h" | tr -d '( ' | sort -u > "$(dirname "$0")/../../tests/golden/mbelib_api_symbols.txt"
wc -l "$(dirname "$0")/../../tests/golden/mbelib_api_symbols.txt"
