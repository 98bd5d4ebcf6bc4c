/* filter_expr.h -- the expression language of `-f` (site filter on AC/AN) and `-s` (sample selection on
 * .spl metadata).  Same entry points and evaluation rules as the reference's kexpr.h/kexpr.c, because the
 * typing rules decide which sites pass: every value carries an integer and a real
 * view; `/` always yields a real, `//` an integer; comparisons are made on the reals if either side is
 * real; a variable that was never bound makes the evaluation report an error. */
#ifndef BGT_FILTER_EXPR_H
#define BGT_FILTER_EXPR_H
#include <stdint.h>

typedef struct kexpr_s kexpr_t;

#define KEE_UNQU   0x01
#define KEE_UNLP   0x02
#define KEE_UNRP   0x04
#define KEE_UNOP   0x08
#define KEE_FUNC   0x10
#define KEE_ARG    0x20
#define KEE_NUM    0x40
#define KEE_UNFUNC 0x40
#define KEE_UNVAR  0x80

#define KEV_REAL 1
#define KEV_INT  2
#define KEV_STR  3

#ifdef __cplusplus
extern "C" {
#endif
kexpr_t *ke_parse(const char *s, int *err);
void     ke_destroy(kexpr_t *ke);
kexpr_t *ke_clone(const kexpr_t *ke);                /* extension: an independent copy (bindings included) */
int      ke_set_int(kexpr_t *ke, const char *var, int64_t x);
int      ke_set_real(kexpr_t *ke, const char *var, double x);
int      ke_set_str(kexpr_t *ke, const char *var, const char *x);
void     ke_unset(kexpr_t *ke);
int      ke_eval(const kexpr_t *ke, int64_t *i, double *r, const char **s, int *ret_type);
int64_t  ke_eval_int(const kexpr_t *ke, int *err);
double   ke_eval_real(const kexpr_t *ke, int *err);
/* Compact program for the device-side filter: returns the number of RPN items, fills parallel arrays
 * (op: 0 = push int const, 1 = push real const, 2 = push variable #arg, >=16 = operator code - 16). */
int      ke_export(const kexpr_t *ke, int max_items, int32_t *op, int64_t *ival, double *rval,
                   const char **var_name);
#ifdef __cplusplus
}
#endif
#endif
