/* filter_expr.c -- see filter_expr.h.  Shunting-yard to reverse Polish, then a stack machine in which every
 * slot holds {int64 i, double r, string s, type}.  Restates reference kexpr.c:14-153 (operators and their
 * typing), :155-243 (tokens), :255-355 (precedence climbing), :366-418 (evaluation). */
#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "filter_expr.h"

enum { T_VAL = 1, T_OP, T_FUNC, T_LPAR };
enum { O_POS = 1, O_NEG, O_BNOT, O_LNOT, O_POW, O_MUL, O_DIV, O_IDIV, O_MOD, O_ADD, O_SUB, O_LSH, O_RSH,
       O_LT, O_LE, O_GT, O_GE, O_EQ, O_NE, O_BAND, O_BXOR, O_BOR, O_LAND, O_LOR };

/* precedence level (smaller binds tighter) and right-associativity, ref kexpr.c:66-80 */
static const struct { int level, right; } prec[25] = {
    {0, 0}, {1, 1}, {1, 1}, {1, 1}, {1, 1}, {2, 1}, {3, 0}, {3, 0}, {3, 0}, {3, 0}, {4, 0}, {4, 0}, {5, 0}, {5, 0},
    {6, 0}, {6, 0}, {6, 0}, {6, 0}, {7, 0}, {7, 0}, {8, 0}, {9, 0}, {10, 0}, {11, 0}, {12, 0}};

typedef struct {
    int kind, op, n_args, vtype, assigned, fn;      /* fn: 0 none, 1 abs */
    char *name, *s;
    int64_t i;
    double r;
} item_t;

struct kexpr_s { int n; item_t *e; };

typedef struct { item_t *a; int n, m; } vec_t;
static item_t *vpush(vec_t *v)
{
    if (v->n == v->m) { v->m = v->m ? v->m * 2 : 8; v->a = (item_t*)realloc(v->a, (size_t)v->m * sizeof(item_t)); }
    memset(&v->a[v->n], 0, sizeof(item_t));
    return &v->a[v->n++];
}
static char *dupn(const char *s, size_t n) { char *d = (char*)calloc(n + 1, 1); memcpy(d, s, n); return d; }

static int two(const char *p, char a, char b) { return p[0] == a && p[1] == b; }

/* one token that is not a parenthesis or a comma */
static item_t read_token(const char *p, const char **end, int *err, int last_is_val)
{
    item_t e;
    const char *q = p;
    memset(&e, 0, sizeof(e));
    if (isalpha((unsigned char)*p) || *p == '_') {
        while (*p && (*p == '_' || isalnum((unsigned char)*p))) ++p;
        if (*p == '(') { e.kind = T_FUNC; e.n_args = 1; }
        else { e.kind = T_VAL; e.vtype = KEV_REAL; }
        e.name = dupn(q, (size_t)(p - q));
        *end = p;
    } else if (isdigit((unsigned char)*p) || *p == '.') {
        char *pr, *pi;
        double y = strtod(q, &pr);
        long x = strtol(q, &pi, 0);
        e.kind = T_VAL;
        if (pr == q && pi == q) *err |= KEE_NUM;
        else if (pr > pi) { e.vtype = KEV_REAL; e.i = (int64_t)(y + .5); e.r = y; *end = pr; }
        else { e.vtype = KEV_INT; e.i = x; e.r = y; *end = pi; }
    } else if (*p == '"' || *p == '\'') {
        int c = *p;
        for (++p; *p && *p != c; ++p) if (*p == '\\') ++p;
        if (*p == c) { e.kind = T_VAL; e.vtype = KEV_STR; e.s = dupn(q + 1, (size_t)(p - q - 1)); *end = p + 1; }
        else { *err |= KEE_UNQU; *end = p; }
    } else {
        int len = 1;
        e.kind = T_OP; e.n_args = 2;
        if (two(p, '*', '*')) e.op = O_POW, len = 2;
        else if (*p == '*') e.op = O_MUL;
        else if (two(p, '/', '/')) e.op = O_IDIV, len = 2;
        else if (*p == '/') e.op = O_DIV;
        else if (*p == '%') e.op = O_MOD;
        else if (*p == '+') { if (last_is_val) e.op = O_ADD; else e.op = O_POS, e.n_args = 1; }
        else if (*p == '-') { if (last_is_val) e.op = O_SUB; else e.op = O_NEG, e.n_args = 1; }
        else if (two(p, '=', '=')) e.op = O_EQ, len = 2;
        else if (two(p, '!', '=') || two(p, '<', '>')) e.op = O_NE, len = 2;
        else if (two(p, '>', '=')) e.op = O_GE, len = 2;
        else if (two(p, '<', '=')) e.op = O_LE, len = 2;
        else if (two(p, '>', '>')) e.op = O_RSH, len = 2;
        else if (two(p, '<', '<')) e.op = O_LSH, len = 2;
        else if (*p == '>') e.op = O_GT;
        else if (*p == '<') e.op = O_LT;
        else if (two(p, '|', '|')) e.op = O_LOR, len = 2;
        else if (two(p, '&', '&')) e.op = O_LAND, len = 2;
        else if (*p == '|') e.op = O_BOR;
        else if (*p == '&') e.op = O_BAND;
        else if (*p == '^') e.op = O_BXOR;
        else if (*p == '~') e.op = O_BNOT, e.n_args = 1;
        else if (*p == '!') e.op = O_LNOT, e.n_args = 1;
        else { e.kind = 0; *err |= KEE_UNOP; }
        *end = q + len;
    }
    return e;
}

static void free_items(item_t *a, int n) { int i; for (i = 0; i < n; ++i) { free(a[i].name); free(a[i].s); } free(a); }

kexpr_t *ke_parse(const char *src, int *err)
{
    char *s = (char*)malloc(strlen(src) + 1), *w = s;
    const char *p;
    vec_t out = {0, 0, 0}, ops = {0, 0, 0};
    int last_is_val = 0, i;
    kexpr_t *ke;
    *err = 0;
    for (p = src; *p; ++p) if (!isspace((unsigned char)*p)) *w++ = *p;   /* blanks are dropped everywhere */
    *w = 0;
    p = s;
    while (*p) {
        if (*p == '(') { item_t *t = vpush(&ops); t->kind = T_LPAR; ++p; }
        else if (*p == ')') {
            while (ops.n > 0 && ops.a[ops.n - 1].kind != T_LPAR) *vpush(&out) = ops.a[--ops.n];
            if (ops.n == 0) { *err |= KEE_UNRP; break; }
            --ops.n;
            if (ops.n > 0 && ops.a[ops.n - 1].kind == T_FUNC) {
                item_t *u = vpush(&out);
                *u = ops.a[--ops.n];
                if (u->n_args == 1 && strcmp(u->name, "abs") == 0) u->fn = 1;
            }
            ++p;
        } else if (*p == ',') {
            while (ops.n > 0 && ops.a[ops.n - 1].kind != T_LPAR) *vpush(&out) = ops.a[--ops.n];
            if (ops.n < 2 || ops.a[ops.n - 2].kind != T_FUNC) { *err |= KEE_FUNC; break; }
            ++ops.a[ops.n - 2].n_args;
            ++p;
        } else {
            item_t v = read_token(p, &p, err, last_is_val);
            if (*err) { free(v.name); free(v.s); break; }
            if (v.kind == T_VAL) { *vpush(&out) = v; last_is_val = 1; }
            else if (v.kind == T_FUNC) { *vpush(&ops) = v; last_is_val = 0; }
            else {
                while (ops.n > 0 && ops.a[ops.n - 1].kind == T_OP) {
                    const int top = prec[ops.a[ops.n - 1].op].level;
                    if ((prec[v.op].right && prec[v.op].level <= top) || (!prec[v.op].right && prec[v.op].level < top)) break;
                    *vpush(&out) = ops.a[--ops.n];
                }
                *vpush(&ops) = v;
                last_is_val = 0;
            }
        }
    }
    if (*err == 0) {
        while (ops.n > 0 && ops.a[ops.n - 1].kind != T_LPAR) *vpush(&out) = ops.a[--ops.n];
        if (ops.n > 0) *err |= KEE_UNLP;
    }
    if (*err == 0) {                                   /* the program must leave exactly one value */
        int depth = 0;
        for (i = 0; i < out.n; ++i) depth += out.a[i].kind == T_VAL ? 1 : -(out.a[i].n_args - 1);
        if (depth != 1) *err |= KEE_ARG;
    }
    free(s);
    if (*err) {
        for (i = 0; i < ops.n; ++i) { free(ops.a[i].name); free(ops.a[i].s); }
        free(ops.a); free_items(out.a, out.n);
        return NULL;
    }
    free(ops.a);
    ke = (kexpr_t*)calloc(1, sizeof(*ke));
    ke->n = out.n; ke->e = out.a;
    return ke;
}

void ke_destroy(kexpr_t *ke) { if (ke) { free_items(ke->e, ke->n); free(ke); } }

/* an independent copy (an expression carries its variable bindings: one per thread that evaluates it) */
kexpr_t *ke_clone(const kexpr_t *ke)
{
    kexpr_t *c;
    int i;
    if (ke == NULL) return NULL;
    c = (kexpr_t*)calloc(1, sizeof(*c));
    c->n = ke->n;
    c->e = (item_t*)malloc((size_t)(ke->n ? ke->n : 1) * sizeof(item_t));
    memcpy(c->e, ke->e, (size_t)ke->n * sizeof(item_t));
    for (i = 0; i < ke->n; ++i) {
        if (ke->e[i].name) c->e[i].name = dupn(ke->e[i].name, strlen(ke->e[i].name));
        if (ke->e[i].s) c->e[i].s = dupn(ke->e[i].s, strlen(ke->e[i].s));
    }
    return c;
}

#define FOR_VAR(ke, var, body) do { int i_, n_ = 0; for (i_ = 0; i_ < (ke)->n; ++i_) { item_t *e = &(ke)->e[i_]; \
    if (e->kind == T_VAL && e->name && strcmp(e->name, (var)) == 0) { body; ++n_; } } return n_; } while (0)

int ke_set_int(kexpr_t *ke, const char *var, int64_t x)
{ FOR_VAR(ke, var, (e->i = x, e->r = (double)x, e->vtype = KEV_INT, e->assigned = 1)); }
int ke_set_real(kexpr_t *ke, const char *var, double x)
{ FOR_VAR(ke, var, (e->r = x, e->i = (int64_t)(x + .5), e->vtype = KEV_REAL, e->assigned = 1)); }
int ke_set_str(kexpr_t *ke, const char *var, const char *x)
{ FOR_VAR(ke, var, (free(e->s), e->s = dupn(x, strlen(x)), e->i = 0, e->r = 0., e->vtype = KEV_STR, e->assigned = 1)); }

void ke_unset(kexpr_t *ke) { int i; for (i = 0; i < ke->n; ++i) if (ke->e[i].kind == T_VAL && ke->e[i].name) ke->e[i].assigned = 0; }

typedef struct { int64_t i; double r; const char *s; int t; } slot_t;

static void apply2(int op, slot_t *p, const slot_t *q)
{
    const int real = p->t == KEV_REAL || q->t == KEV_REAL;
    int c;
    switch (op) {
    case O_LT: case O_LE: case O_GT: case O_GE: case O_EQ: case O_NE:
        if (p->t == KEV_STR && q->t == KEV_STR) { int d = strcmp(p->s, q->s);
            c = op == O_LT ? d < 0 : op == O_LE ? d <= 0 : op == O_GT ? d > 0 : op == O_GE ? d >= 0 : op == O_EQ ? d == 0 : d != 0; }
        else if (real) c = op == O_LT ? p->r < q->r : op == O_LE ? p->r <= q->r : op == O_GT ? p->r > q->r :
                           op == O_GE ? p->r >= q->r : op == O_EQ ? p->r == q->r : p->r != q->r;
        else c = op == O_LT ? p->i < q->i : op == O_LE ? p->i <= q->i : op == O_GT ? p->i > q->i :
                 op == O_GE ? p->i >= q->i : op == O_EQ ? p->i == q->i : p->i != q->i;
        p->i = c; p->r = (double)c; p->t = KEV_INT; break;
    case O_BAND: p->i &= q->i; goto int_done;
    case O_BOR:  p->i |= q->i; goto int_done;
    case O_BXOR: p->i ^= q->i; goto int_done;
    case O_LSH:  p->i <<= q->i; goto int_done;
    case O_RSH:  p->i >>= q->i; goto int_done;
    case O_MOD:  p->i %= q->i; goto int_done;          /* a zero divisor traps, as in the reference */
    case O_IDIV: p->i /= q->i;
    int_done:    p->r = (double)p->i; p->t = KEV_INT; break;
    case O_ADD: p->i += q->i; p->r += q->r; p->t = real ? KEV_REAL : KEV_INT; break;
    case O_SUB: p->i -= q->i; p->r -= q->r; p->t = real ? KEV_REAL : KEV_INT; break;
    case O_MUL: p->i *= q->i; p->r *= q->r; p->t = real ? KEV_REAL : KEV_INT; break;
    case O_DIV: p->r /= q->r; p->i = (int64_t)(p->r + .5); p->t = KEV_REAL; break;
    case O_POW: p->r = pow(p->r, q->r); p->i = (int64_t)(p->r + .5); p->t = real ? KEV_REAL : KEV_INT; break;
    case O_LAND: p->i = (p->i && q->i); p->r = (double)p->i; p->t = KEV_INT; break;
    case O_LOR:  p->i = (p->i || q->i); p->r = (double)p->i; p->t = KEV_INT; break;
    }
}

static void apply1(int op, slot_t *p)
{
    switch (op) {
    case O_NEG:  p->i = -p->i; p->r = -p->r; break;
    case O_BNOT: p->i = ~p->i; p->r = (double)p->i; p->t = KEV_INT; break;
    case O_LNOT: p->i = !p->i; p->r = (double)p->i; p->t = KEV_INT; break;
    default: break;                                     /* unary plus */
    }
}

int ke_eval(const kexpr_t *ke, int64_t *oi, double *orr, const char **os, int *ret_type)
{
    slot_t *st = (slot_t*)malloc((size_t)(ke->n > 0 ? ke->n : 1) * sizeof(slot_t));
    int i, top = 0, err = 0;
    *oi = 0; *orr = 0.; *ret_type = 0;
    for (i = 0; i < ke->n; ++i) {
        const item_t *e = &ke->e[i];
        if (e->kind == T_FUNC && !e->fn) err |= KEE_UNFUNC;         /* only abs() is built in */
        else if (e->kind == T_VAL && e->name && !e->assigned) err |= KEE_UNVAR;
    }
    for (i = 0; i < ke->n; ++i) {
        const item_t *e = &ke->e[i];
        if (e->kind == T_VAL) { st[top].i = e->i; st[top].r = e->r; st[top].s = e->s; st[top].t = e->vtype; ++top; }
        else if (e->kind == T_OP) {
            if (e->n_args == 2) { --top; apply2(e->op, &st[top - 1], &st[top]); }
            else apply1(e->op, &st[top - 1]);
        } else if (e->fn == 1) {                                     /* abs */
            slot_t *p = &st[top - 1];
            if (p->t == KEV_INT) { p->i = (int64_t)abs((int)p->i); p->r = (double)p->i; }   /* int abs(), ref kexpr.c:153 */
            else { p->r = fabs(p->r); p->i = (int64_t)(p->r + .5); }
        } else top -= e->n_args - 1;                                 /* unknown function: drop its arguments */
    }
    *ret_type = st[0].t; *oi = st[0].i; *orr = st[0].r; *os = st[0].s;
    free(st);
    return err;
}

int64_t ke_eval_int(const kexpr_t *ke, int *err) { int64_t i; double r; const char *s; int t; *err = ke_eval(ke, &i, &r, &s, &t); return i; }
double ke_eval_real(const kexpr_t *ke, int *err) { int64_t i; double r; const char *s; int t; *err = ke_eval(ke, &i, &r, &s, &t); return r; }

int ke_export(const kexpr_t *ke, int max_items, int32_t *op, int64_t *ival, double *rval, const char **var_name)
{
    int i;
    if (ke->n > max_items) return -1;
    for (i = 0; i < ke->n; ++i) {
        const item_t *e = &ke->e[i];
        ival[i] = e->i; rval[i] = e->r; var_name[i] = e->name;
        if (e->kind == T_VAL) {
            if (e->vtype == KEV_STR && !e->name) return -2;          /* strings stay on the host */
            op[i] = e->name ? 2 : e->vtype == KEV_INT ? 0 : 1;
        } else if (e->kind == T_OP) op[i] = 16 + e->op;
        else return -2;
    }
    return ke->n;
}
