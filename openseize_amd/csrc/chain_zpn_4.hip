// The zero-phase chain kernel on one real block per transform (chain_zpn_body.h) for cascades of
// 4 modes: every block height (24 .. 30 rows) and slow-mode count.
#define OSZ_ZPN_NM 4
#include "chain_zpn_body.h"
