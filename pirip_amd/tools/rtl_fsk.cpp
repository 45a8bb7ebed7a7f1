// rtl_fsk -- placeholder main; the full argv surface is built in rtl_fsk once section A is
// parity-green.
#include <cstdio>
int main() { fprintf(stderr, "rtl_fsk: not built yet\n"); return 2; }
