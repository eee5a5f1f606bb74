/* Source-compatibility shim: callers of the reference include "recur-nn.h"
 * (e.g. text-predict.c:12, gstclassify.h, charmodel.h:3).  Pointing their
 * include path at this directory gives them the same structs and prototypes,
 * implemented by librecur_amd.so.  See INTEGRATION.md. */
#ifndef _GOT_RECUR_NN_H
#define _GOT_RECUR_NN_H 1
#include <unistd.h>
#include <string.h>
#include <stdlib.h>
#include <inttypes.h>
#include "recur_amd.h"
#endif
