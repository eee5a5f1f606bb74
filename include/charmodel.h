/* charmodel.h -- lets the reference's text tools (text-predict.c, text-cross-entropy.c,
 * text-confabulate.c) compile against librecur_amd unchanged: the declarations they
 * use from the reference's charmodel.h live in recur_amd_char.h. */
#ifndef HAVE_CHAR_MODEL_H
#define HAVE_CHAR_MODEL_H
#include "recur_amd_char.h"
#endif
