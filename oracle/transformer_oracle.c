/* transformer_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE). Filled in below. */
#include "bitnet_oracle.h"
