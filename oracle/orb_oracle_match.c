/* placeholder translation unit: matcher restatements are added in orb_oracle_match.c */
#include "orb_oracle.h"
