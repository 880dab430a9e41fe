/*
 * oracle/pm_oracle.c -- TEST INFRASTRUCTURE (oracle). Not part of the product.
 * See pm_oracle.h.  Every function cites the reference lines it restates.
 * Build with -ffp-contract=off (the reference is plain IEEE double).
 */
#include "pm_oracle.h"
#include "orc_rng.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tape */
void orc_tape_init(orc_tape *t) { t->v = NULL; t->n = 0; t->cap = 0; }
void orc_tape_free(orc_tape *t) { free(t->v); t->v = NULL; t->n = t->cap = 0; }
static void tape_push(orc_tape *t, double v) {
    if (!t) return;
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->v = (double *)realloc(t->v, t->cap * sizeof(double));
    }
    t->v[t->n++] = v;
}
static double draw(uint64_t IK, uint32_t j, orc_tape *t) {
    double v = orc_draw(IK, j);
    tape_push(t, v);
    return v;
}

/* --------------------------------------------------------------- helpers */
/* src/Vector.cpp:13-16 (x86 cvttsd2si gives INT_MIN on NaN / overflow) */
int32_t orc_fastfloor(double x) {
    int32_t xi;
    if (!(x > -2147483649.0 && x < 2147483648.0)) xi = INT32_MIN; else xi = (int32_t)x;
    return x < xi ? xi - 1 : xi;
}
/* src/Vector.cpp:18-20 */
static int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
/* src/Vector.cpp:95-100 : three divisions by the length */
static void normalize3(double v[3]) {
    double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= len; v[1] /= len; v[2] /= len;
}

/* src/Image.cpp:20-28 ; colour literal on line 22 truncates to EE,EE,EE */
static void set_material(orc_prim *p, uint32_t col, double d, double s, double t, double a) {
    p->red = ((col >> 16) & 0xFF) / 255.0f;
    p->green = ((col >> 8) & 0xFF) / 255.0f;
    p->blue = (col & 0xFF) / 255.0f;
    p->absorb_chance = a; p->diffuse_chance = d; p->specular_chance = s; p->transmission_chance = t;
    p->refraction = 0; p->specular_coeff = 0;
}
static void make_plane(orc_prim *p, double x, double y, double z, double mn, double mx, uint32_t col,
                       double d, double s, double t, double a) {
    memset(p, 0, sizeof(*p));
    p->kind = 1; p->x = x; p->y = y; p->z = z; p->min_bound = mn; p->max_bound = mx;
    set_material(p, col, d, s, t, a);
}
static void make_sphere(orc_prim *p, double x, double y, double z, double r, uint32_t col, double d,
                        double s, double t, double a) {
    memset(p, 0, sizeof(*p));
    p->kind = 0; p->x = x; p->y = y; p->z = z; p->radius = r;
    set_material(p, col, d, s, t, a);
}
void orc_default_scene(orc_prim o[7]) {
    make_plane(&o[0], 0, -5, 0, -5, 5, 0xFFEEEEEE, 0.4, 0.0, 0.0, 1.0);
    make_plane(&o[1], 0, 5, 0, -5, 5, 0xFDEEEEEE, 0.4, 0.0, 0.0, 1.0);
    make_plane(&o[2], 5, 0, 0, -5, 5, 0xFFFF3333, 0.4, 0.0, 0.0, 1.0);
    make_plane(&o[3], -5, 0, 0, -5, 5, 0xFF3333FF, 0.4, 0.0, 0.0, 1.0);
    make_plane(&o[4], 0, 0, 10, -5, 5, 0xFFEEEEEE, 0.4, 0.0, 0.0, 1.0);
    make_sphere(&o[5], 2, -3.5, 3, 1.5, 0xFFFFFFFF, 0.0, 0.1, 0.9, 0.0);
    o[5].refraction = 2.5;
    make_sphere(&o[6], -2, -3.5, 5, 1.5, 0xFFFFFFFF, 0.0, 1.0, 0.0, 0.0);
}

/* ------------------------------------------------------------- intersect */
/* src/Scene.cpp:87-116 */
static int sphere_intersect(const orc_prim *s, const double o[3], const double d[3], double dt,
                            double res[3], double nrm[3]) {
    double x0 = s->x + dt * s->dx;
    double y0 = s->y + dt * s->dy;
    double z0 = s->z + dt * s->dz;
    double lx = x0 - o[0], ly = y0 - o[1], lz = z0 - o[2];
    double b = d[0] * lx + d[1] * ly + d[2] * lz;
    if (b < 0) return 0;
    double d2 = (lx * lx + ly * ly + lz * lz) - b * b;
    if (d2 > s->radius * s->radius) return 0;
    double t = sqrt(s->radius * s->radius - d2);
    if (t < 0) t = b + t; else t = b - t;
    if (t < 0) return 0;
    res[0] = d[0] * t + o[0]; res[1] = d[1] * t + o[1]; res[2] = d[2] * t + o[2];
    nrm[0] = res[0] - x0; nrm[1] = res[1] - y0; nrm[2] = res[2] - z0;
    normalize3(nrm);
    return 1;
}
/* src/Scene.cpp:135-193 */
static int plane_intersect(const orc_prim *p, const double c[3], const double r[3], double res[3],
                           double nrm[3]) {
    if (p->x != 0) {
        double dx = p->x - c[0];
        double mul = dx / r[0];
        if (mul < 0) return 0;
        double hz = c[2] + mul * r[2];
        if (hz < -0.01) return 0;
        double hy = c[1] + mul * r[1];
        if (hy > p->max_bound || hy < p->min_bound) return 0;
        res[0] = p->x; res[1] = hy; res[2] = hz;
        nrm[0] = p->x < 0 ? 1 : -1; nrm[1] = 0; nrm[2] = 0;
        return 1;
    } else if (p->y != 0) {
        double dy = p->y - c[1];
        double mul = dy / r[1];
        if (mul < 0) return 0;
        double hz = c[2] + mul * r[2];
        if (hz < -0.01) return 0;
        double hx = c[0] + mul * r[0];
        if (hx < p->min_bound || hx > p->max_bound) return 0;
        res[0] = hx; res[1] = p->y; res[2] = hz;
        nrm[0] = 0; nrm[1] = p->y < 0 ? 1 : -1; nrm[2] = 0;
        return 1;
    } else if (p->z > 0) {
        double dz = p->z - c[2];
        double mul = dz / r[2];
        if (mul < 0) return 0;
        double hx = c[0] + mul * r[0];
        if (hx < -5 || hx > 5) return 0;
        double hy = c[1] + mul * r[1];
        if (hy < p->min_bound || hy > p->max_bound) return 0;
        res[0] = hx; res[1] = hy; res[2] = p->z;
        nrm[0] = 0; nrm[1] = 0; nrm[2] = p->z < 0 ? 1 : -1;
        return 1;
    }
    return 0;
}
int orc_prim_intersect(const orc_prim *p, const double src[3], const double dir[3], double dt,
                       double res[3], double nrm[3]) {
    return p->kind == 0 ? sphere_intersect(p, src, dir, dt, res, nrm)
                        : plane_intersect(p, src, dir, res, nrm);
}
/* src/Scene.cpp:25-47 : returns object index or -1 */
int orc_scene_intersect(const orc_prim *prims, int n, const double src[3], const double dir[3],
                        int exclude, double dt, double fres[3], double fnrm[3]) {
    double nearest = 1024 * 1024;
    int nearest_obj = -1;
    double res[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        if (i == exclude) continue;
        if (orc_prim_intersect(&prims[i], src, dir, dt, res, nrm)) {
            double dist = (res[0] - src[0]) * (res[0] - src[0]);
            dist += (res[1] - src[1]) * (res[1] - src[1]);
            dist += (res[2] - src[2]) * (res[2] - src[2]);
            if (dist < nearest) {
                nearest = dist; nearest_obj = i;
                fres[0] = res[0]; fres[1] = res[1]; fres[2] = res[2];
                fnrm[0] = nrm[0]; fnrm[1] = nrm[1]; fnrm[2] = nrm[2];
            }
        }
    }
    return nearest_obj;
}

/* ------------------------------------------------- deterministic sin/cos */
/* For psi in [0, 2*pi): quadrant reduction + fdlibm-style kernels, fixed
 * operation order, no FMA.  Used (ORC_TRIG_DET) so that the GPU emission can be
 * compared bit for bit; ORC_TRIG_LIBM calls glibc like the reference does
 * (src/PhotonMap.cpp:299,412).  tests/ bound the difference between the two. */
void orc_det_sincos(double x, double *sn, double *cs) {
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    static const double INVPIO2 = 6.36619772367581382433e-01;
    static const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    static const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    int q = (int)(x * INVPIO2 + 0.5);
    double fq = (double)q;
    double r = (x - fq * PIO2_1) - fq * PIO2_1T;
    double z = r * r;
    /* sin kernel */
    double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double s = r + (z * r) * (S1 + z * ps);
    /* cos kernel */
    double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double c = 1.0 - (0.5 * z - z * pc);
    switch (q & 3) {
    case 0: *sn = s; *cs = c; break;
    case 1: *sn = c; *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
    }
}

/* ---------------------------------------------------------- primary rays */
/* src/Raytrace.cpp:284-293 */
void orc_primary_ray(int width, int height, const double cam[3], uint64_t seed, int x, int y,
                     double dir[3]) {
    uint64_t K = orc_stream_key(seed, ORC_KIND_PIXEL);
    uint64_t pixel = (uint64_t)x + (uint64_t)y * (uint64_t)width;
    uint64_t IK0 = orc_item_key(K, pixel * 32u);
    double fov = (width / 1280.0) * 64.0;
    double x1 = orc_draw(IK0, 0) * 0.6 - 0.3;
    double y1 = orc_draw(IK0, 1) * 0.6 - 0.3;
    double x0 = (x - width / 2 + x1) / fov - cam[0];
    double y0 = (y - height / 2 + y1) / fov - cam[1];
    dir[0] = x0; dir[1] = y0; dir[2] = -cam[2];
    normalize3(dir);
}
void orc_primary_hits(const orc_prim *prims, int n, int width, int height, const double cam[3],
                      uint64_t seed, int y0, int y1, int32_t *ids, double *hit, double *nrm) {
    uint64_t K = orc_stream_key(seed, ORC_KIND_PIXEL);
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < width; x++) {
            double dir[3], res[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
            orc_primary_ray(width, height, cam, seed, x, y, dir);
            uint64_t pixel = (uint64_t)x + (uint64_t)y * (uint64_t)width;
            double dt = orc_draw(orc_item_key(K, pixel * 32u + 1u), 0);
            int id = orc_scene_intersect(prims, n, cam, dir, -1, dt, res, nn);
            size_t o = (size_t)(y - y0) * width + x;
            ids[o] = id;
            if (hit) { hit[3 * o] = res[0]; hit[3 * o + 1] = res[1]; hit[3 * o + 2] = res[2]; }
            if (nrm) { nrm[3 * o] = nn[0]; nrm[3 * o + 1] = nn[1]; nrm[3 * o + 2] = nn[2]; }
        }
}

/* -------------------------------------------------------------- emission */
/* One pass of the outer while-loop body of createPhotonMap (src/PhotonMap.cpp:288-367)
 * or createCausticPhotonMap (:400-484). */
int orc_emit_attempt(const orc_prim *prims, int n, int map_kind, uint64_t seed, uint64_t attempt,
                     int trig, const double light_color[3], orc_photon *out, int *ndraws,
                     orc_tape *tape) {
    uint64_t K = orc_stream_key(seed, map_kind == ORC_MAP_GLOBAL ? ORC_KIND_EMIT_GLOBAL
                                                               : ORC_KIND_EMIT_CAUSTIC);
    uint64_t IK = orc_item_key(K, attempt);
    uint32_t j = 0;
    double power[3] = {light_color[0], light_color[1], light_color[2]};
    double x0 = draw(IK, j++, tape) * 2 - 1;
    double z0 = draw(IK, j++, tape) * 2 + 3;
    double y0 = 4.95;
    double src[3] = {x0, y0, z0};
    double sin_theta = sqrt(draw(IK, j++, tape));
    double cos_theta = sqrt(1 - sin_theta * sin_theta);
    double psi = draw(IK, j++, tape) * 6.2831853;
    double sp, cp;
    if (trig == ORC_TRIG_LIBM) { cp = cos(psi); sp = sin(psi); } else orc_det_sincos(psi, &sp, &cp);
    double dir[3];
    dir[0] = sin_theta * cp;
    dir[1] = map_kind == ORC_MAP_GLOBAL ? -cos_theta : -2 * cos_theta;
    dir[2] = sin_theta * sp;
    normalize3(dir);
    int bounces = 0, exclude = -1, specular_bounce = 0, stored = 0;
    double res[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
    for (;;) {
        bounces++;
        int obj = orc_scene_intersect(prims, n, src, dir, exclude, 0, res, nrm);
        if (obj < 0) break;
        const orc_prim *o = &prims[obj];
        double chance = draw(IK, j++, tape);
        if (bounces > 3) chance = 1;
        if (chance < o->diffuse_chance + o->specular_chance) {
            double n1[3] = {nrm[0], nrm[1], nrm[2]};
            double k = n1[0] * dir[0] + n1[1] * dir[1] + n1[2] * dir[2];
            n1[0] *= k; n1[1] *= k; n1[2] *= k;
            n1[0] *= 2; n1[1] *= 2; n1[2] *= 2;
            src[0] = res[0]; src[1] = res[1]; src[2] = res[2];
            dir[0] = dir[0] - n1[0]; dir[1] = dir[1] - n1[1]; dir[2] = dir[2] - n1[2];
            normalize3(dir);
            power[0] *= o->red; power[1] *= o->green; power[2] *= o->blue;
            if (map_kind == ORC_MAP_GLOBAL) exclude = obj; /* :327 ; caustic passes nullptr (:420) */
            specular_bounce = 1;
            continue;
        } else if (chance < o->diffuse_chance + o->specular_chance + o->transmission_chance) {
            double nn = 1 / o->refraction;
            double d = nrm[0] * dir[0] + nrm[1] * dir[1] + nrm[2] * dir[2];
            double n1[3] = {nrm[0] * d, nrm[1] * d, nrm[2] * d};
            n1[0] = dir[0] - n1[0]; n1[1] = dir[1] - n1[1]; n1[2] = dir[2] - n1[2];
            n1[0] *= nn; n1[1] *= nn; n1[2] *= nn;
            double s = 1 - (nn * nn) * (1 - d * d);
            double sq = sqrt(s);
            double n2[3] = {nrm[0] * sq, nrm[1] * sq, nrm[2] * sq};
            n1[0] += -n2[0]; n1[1] += -n2[1]; n1[2] += -n2[2];
            src[0] = res[0] + n1[0] * 0.01; src[1] = res[1] + n1[1] * 0.01; src[2] = res[2] + n1[2] * 0.01;
            dir[0] = n1[0]; dir[1] = n1[1]; dir[2] = n1[2];
            normalize3(dir);
            exclude = -1;
            specular_bounce = 1;
            continue;
        } else {
            if (map_kind == ORC_MAP_CAUSTIC && !specular_bounce) break; /* :463-466 */
            memset(out, 0, sizeof(*out));
            out->x = (float)res[0]; out->y = (float)res[1]; out->z = (float)res[2];
            out->power[0] = (uint8_t)orc_fastfloor(power[0] * 0xFF);
            out->power[1] = (uint8_t)orc_fastfloor(power[1] * 0xFF);
            out->power[2] = (uint8_t)orc_fastfloor(power[2] * 0xFF);
            out->power[3] = 255;
            out->dx = (float)dir[0]; out->dy = (float)dir[1]; out->dz = (float)dir[2];
            out->bounce = (int8_t)bounces;
            stored = 1;
        }
        break;
    }
    if (ndraws) *ndraws = (int)j;
    return stored;
}
uint64_t orc_emit(const orc_prim *prims, int n, int map_kind, uint64_t seed, int n_photons, int trig,
                  const double light_color[3], orc_photon *out, orc_tape *tape) {
    uint64_t attempt = 0;
    int stored = 0;
    while (stored < n_photons) {
        if (orc_emit_attempt(prims, n, map_kind, seed, attempt, trig, light_color, &out[stored], NULL, tape))
            stored++;
        attempt++;
    }
    return attempt;
}

/* ------------------------------------------------------------ photon map */
typedef struct refnode { int32_t photon, left, right, axis; } refnode; /* kdnode, src/PhotonMap.h:15-20 */
typedef struct exnode { float split; int32_t axis; int32_t lo, hi; int32_t left, right; } exnode;

struct orc_map {
    int n;
    orc_photon *ph;
    /* reference-shaped tree */
    refnode *rn; int rn_count; int rn_root; int ref_depth;
    /* exact balanced kd-tree over idx[] */
    int32_t *idx; exnode *en; int en_count;
};

uint64_t orc_pivot_next(uint64_t pivot_seed, uint64_t *counter, int32_t lo, int32_t hi) {
    uint64_t K = orc_stream_key(pivot_seed, ORC_KIND_KD_PIVOT);
    uint64_t u = orc_draw_u64(orc_item_key(K, *counter), 0);
    (*counter)++;
    return (uint64_t)lo + u % (uint64_t)(hi - lo);
}

static double axis_val(const orc_photon *p, int axis) {
    return axis == 0 ? p->x : (axis == 1 ? p->y : p->z);
}
/* src/Vector.cpp:118-132 */
static int largest_axis(double x, double y, double z) {
    if (x > y) return x > z ? 0 : 2;
    return y > z ? 1 : 2;
}
typedef struct refbuild { orc_map *m; uint64_t seed, counter; } refbuild;

/* src/PhotonMap.cpp:13-61 (not a true quickselect: SURVEY F3) */
static int32_t ref_find_median(refbuild *b, int32_t *photons, int size, int axis) {
    int k = (int)orc_pivot_next(b->seed, &b->counter, 0, size);
    int32_t p = photons[k];
    int32_t *scratch = (int32_t *)malloc(sizeof(int32_t) * (size_t)size);
    int less_index = 0, great_index = size - 1;
    double p_val = axis_val(&b->m->ph[p], axis);
    for (int i = 0; i < size; i++) {
        int32_t next = photons[i];
        if (next == p) continue;
        double val = axis_val(&b->m->ph[next], axis);
        if (val <= p_val) scratch[less_index++] = next; else scratch[great_index--] = next;
    }
    int32_t r;
    if (less_index == size / 2) r = p;
    else if (less_index > size / 2) r = ref_find_median(b, scratch, less_index, axis);
    else r = ref_find_median(b, scratch + great_index + 1, size - great_index - 1, axis);
    free(scratch);
    return r;
}
/* src/PhotonMap.cpp:64-140 */
static int32_t ref_create_kdtree(refbuild *b, int32_t *photons, int size, int32_t *scratch, int depth) {
    orc_map *m = b->m;
    if (depth > m->ref_depth) m->ref_depth = depth;
    double mn[3] = {1000, 1000, 1000}, mx[3] = {-1000, -1000, -1000};
    for (int i = 0; i < size; i++) {
        const orc_photon *p = &m->ph[photons[i]];
        if (p->x < mn[0]) mn[0] = p->x;
        if (p->x > mx[0]) mx[0] = p->x;
        if (p->y < mn[1]) mn[1] = p->y;
        if (p->y > mx[1]) mx[1] = p->y;
        if (p->z < mn[2]) mn[2] = p->z;
        if (p->z > mx[2]) mx[2] = p->z;
    }
    int split = largest_axis(mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]);
    int32_t median = ref_find_median(b, photons, size, split);
    double m_val = axis_val(&m->ph[median], split);
    int less_index = 0, greater_index = size - 1;
    for (int i = 0; i < size; i++) {
        int32_t next = photons[i];
        if (next == median) continue;
        double val = axis_val(&m->ph[next], split);
        if (val <= m_val) scratch[less_index++] = next; else scratch[greater_index--] = next;
    }
    int32_t node = m->rn_count++;
    m->rn[node].axis = split;
    m->rn[node].photon = median;
    int32_t l = -1, r = -1;
    if (less_index != 0) l = ref_create_kdtree(b, scratch, less_index, photons, depth + 1);
    if (greater_index < size - 1)
        r = ref_create_kdtree(b, scratch + greater_index + 1, size - (greater_index + 1),
                              photons + (greater_index + 1), depth + 1);
    m->rn[node].left = l; m->rn[node].right = r;
    return node;
}

/* exact index: classic balanced kd-tree, leaf buckets */
#define EX_LEAF 12
static void ex_select(const orc_photon *ph, int32_t *a, int lo, int hi, int k, int axis) {
    /* quickselect so that a[k] is the k-th by (coord, idx); Hoare partition, median-of-3 */
    while (hi - lo > 1) {
        int mid = lo + (hi - lo) / 2;
        int32_t c[3] = {a[lo], a[mid], a[hi - 1]};
        for (int i = 0; i < 2; i++)
            for (int j2 = 0; j2 < 2 - i; j2++) {
                float va = (float)axis_val(&ph[c[j2]], axis), vb = (float)axis_val(&ph[c[j2 + 1]], axis);
                if (va > vb || (va == vb && c[j2] > c[j2 + 1])) { int32_t t = c[j2]; c[j2] = c[j2 + 1]; c[j2 + 1] = t; }
            }
        int32_t pv = c[1];
        float pvv = (float)axis_val(&ph[pv], axis);
        int i = lo, j2 = hi - 1;
        while (i <= j2) {
            for (;;) { float v = (float)axis_val(&ph[a[i]], axis); if (v < pvv || (v == pvv && a[i] < pv)) i++; else break; }
            for (;;) { float v = (float)axis_val(&ph[a[j2]], axis); if (v > pvv || (v == pvv && a[j2] > pv)) j2--; else break; }
            if (i <= j2) { int32_t t = a[i]; a[i] = a[j2]; a[j2] = t; i++; j2--; }
        }
        if (k <= j2) hi = j2 + 1; else if (k >= i) lo = i; else return;
    }
}
static int32_t ex_build(orc_map *m, int lo, int hi) {
    int32_t node = m->en_count++;
    exnode *e = &m->en[node];
    e->lo = lo; e->hi = hi; e->left = e->right = -1; e->axis = -1; e->split = 0;
    if (hi - lo <= EX_LEAF) return node;
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = lo; i < hi; i++) {
        const orc_photon *p = &m->ph[m->idx[i]];
        if (p->x < mn[0]) mn[0] = p->x;
        if (p->x > mx[0]) mx[0] = p->x;
        if (p->y < mn[1]) mn[1] = p->y;
        if (p->y > mx[1]) mx[1] = p->y;
        if (p->z < mn[2]) mn[2] = p->z;
        if (p->z > mx[2]) mx[2] = p->z;
    }
    float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
    int mid = lo + (hi - lo) / 2;
    ex_select(m->ph, m->idx, lo, hi, mid, axis);
    float split = (float)axis_val(&m->ph[m->idx[mid]], axis);
    int32_t l = ex_build(m, lo, mid);
    int32_t r = ex_build(m, mid, hi);
    e = &m->en[node];
    e->axis = axis; e->split = split; e->left = l; e->right = r;
    return node;
}

orc_map *orc_map_create(const orc_photon *photons, int n, int build_ref, uint64_t pivot_seed) {
    orc_map *m = (orc_map *)calloc(1, sizeof(orc_map));
    m->n = n;
    m->ph = (orc_photon *)malloc(sizeof(orc_photon) * (size_t)(n > 0 ? n : 1));
    memcpy(m->ph, photons, sizeof(orc_photon) * (size_t)n);
    m->rn_root = -1;
    if (build_ref && n > 0) {
        m->rn = (refnode *)malloc(sizeof(refnode) * (size_t)n);
        int32_t *a = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        int32_t *s = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        for (int i = 0; i < n; i++) a[i] = i;
        refbuild b = {m, pivot_seed, 0};
        m->rn_root = ref_create_kdtree(&b, a, n, s, 1);
        free(a); free(s);
    }
    m->idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) m->idx[i] = i;
    m->en = (exnode *)malloc(sizeof(exnode) * (size_t)(2 * (n / (EX_LEAF / 2) + 2)));
    if (n > 0) ex_build(m, 0, n);
    return m;
}
void orc_map_destroy(orc_map *m) {
    if (!m) return;
    free(m->ph); free(m->rn); free(m->idx); free(m->en); free(m);
}
int orc_map_size(const orc_map *m) { return m->n; }
const orc_photon *orc_map_photons(const orc_map *m) { return m->ph; }
int orc_map_ref_depth(const orc_map *m) { return m->ref_depth; }

/* src/PhotonMap.cpp:142-206 : array max-heap, sift-down bound is the literal 31 (k must be 63) */
static void ref_insert(int32_t *nearest, double *distances, int k, int size, int32_t next, double dist) {
    if (size < k) {
        int i = size;
        nearest[i] = next; distances[i] = dist;
        while (i != 0) {
            int parent = (i - 1) / 2;
            if (distances[parent] >= distances[i]) break;
            double t = distances[parent]; int32_t tp = nearest[parent];
            distances[parent] = distances[i]; nearest[parent] = nearest[i];
            distances[i] = t; nearest[i] = tp;
            i = parent;
        }
    } else {
        if (dist > distances[0]) return;
        distances[0] = dist; nearest[0] = next;
        int i = 0;
        while (i < 31) {
            if (distances[2 * i + 1] < distances[2 * i + 2]) {
                if (distances[2 * i + 2] > distances[i]) {
                    int parent = i; i = 2 * i + 2;
                    double t = distances[parent]; int32_t tp = nearest[parent];
                    distances[parent] = distances[i]; nearest[parent] = nearest[i];
                    distances[i] = t; nearest[i] = tp;
                } else break;
            } else {
                if (distances[2 * i + 1] > distances[i]) {
                    int parent = i; i = 2 * i + 1;
                    double t = distances[parent]; int32_t tp = nearest[parent];
                    distances[parent] = distances[i]; nearest[parent] = nearest[i];
                    distances[i] = t; nearest[i] = tp;
                } else break;
            }
        }
    }
}
/* the same, for tests of the host API's raytrace::insert */
void orc_heap_insert(int32_t *nearest, double *distances, int k, int size, int32_t next, double dist) { ref_insert(nearest, distances, k, size, next, dist); }
/* src/PhotonMap.cpp:210-255 */
static int ref_find_nearest(const orc_map *m, int32_t *nearest, double *distances, int k, int size,
                            const double t[3], int32_t root, double max_dist) {
    const refnode *nd = &m->rn[root];
    const orc_photon *v = &m->ph[nd->photon];
    double dx = v->x - t[0], dy = v->y - t[1], dz = v->z - t[2];
    double dist = dx * dx + dy * dy + dz * dz;
    if (dist < max_dist) {
        if (size < k) { ref_insert(nearest, distances, k, size, nd->photon, dist); size++; }
        else if (dist < distances[0]) ref_insert(nearest, distances, k, size, nd->photon, dist);
    }
    int side = 1;
    double axis_dist = 0;
    if (nd->axis == 0) { side = t[0] <= v->x; axis_dist = dx * dx; }
    else if (nd->axis == 1) { side = t[1] <= v->y; axis_dist = dy * dy; }
    else { side = t[2] <= v->z; axis_dist = dz * dz; }
    int32_t side_node = side ? nd->left : nd->right;
    if (side_node >= 0) size = ref_find_nearest(m, nearest, distances, k, size, t, side_node, max_dist);
    if (size == 0 || distances[0] > axis_dist) {
        side_node = !side ? nd->left : nd->right;
        if (side_node >= 0) size = ref_find_nearest(m, nearest, distances, k, size, t, side_node, max_dist);
    }
    return size;
}
int orc_knn_ref(const orc_map *m, const double q[3], int32_t idx[63], double d2[63]) {
    for (int i = 0; i < 63; i++) { idx[i] = -1; d2[i] = 0; }
    if (m->rn_root < 0) return 0;
    return ref_find_nearest(m, idx, d2, 63, 0, q, m->rn_root, 100);
}

/* exact search: max-heap on (d2, idx) lexicographic */
typedef struct kheap { int n; double d[63]; int32_t i[63]; } kheap;
static int key_less(double da, int32_t ia, double db, int32_t ib) { return da < db || (da == db && ia < ib); }
static void kheap_offer(kheap *h, double d, int32_t i) {
    if (!(d < 100)) return;
    if (h->n < 63) {
        int c = h->n++;
        h->d[c] = d; h->i[c] = i;
        while (c > 0) {
            int p = (c - 1) / 2;
            if (!key_less(h->d[p], h->i[p], h->d[c], h->i[c])) break;
            double td = h->d[p]; int32_t ti = h->i[p];
            h->d[p] = h->d[c]; h->i[p] = h->i[c]; h->d[c] = td; h->i[c] = ti; c = p;
        }
    } else if (key_less(d, i, h->d[0], h->i[0])) {
        h->d[0] = d; h->i[0] = i;
        int c = 0;
        for (;;) {
            int l = 2 * c + 1, r = l + 1, b = c;
            if (l < 63 && key_less(h->d[b], h->i[b], h->d[l], h->i[l])) b = l;
            if (r < 63 && key_less(h->d[b], h->i[b], h->d[r], h->i[r])) b = r;
            if (b == c) break;
            double td = h->d[b]; int32_t ti = h->i[b];
            h->d[b] = h->d[c]; h->i[b] = h->i[c]; h->d[c] = td; h->i[c] = ti; c = b;
        }
    }
}
static double photon_d2(const orc_photon *v, const double t[3]) {
    double dx = v->x - t[0], dy = v->y - t[1], dz = v->z - t[2];
    return dx * dx + dy * dy + dz * dz;
}
static void ex_search(const orc_map *m, int32_t node, const double q[3], kheap *h) {
    const exnode *e = &m->en[node];
    if (e->left < 0) {
        for (int i = e->lo; i < e->hi; i++) kheap_offer(h, photon_d2(&m->ph[m->idx[i]], q), m->idx[i]);
        return;
    }
    double diff = (double)e->split - q[e->axis];
    int near = diff > 0 ? e->left : e->right, far = diff > 0 ? e->right : e->left;
    ex_search(m, near, q, h);
    double d2 = diff * diff;
    if (h->n < 63 ? d2 < 100 : d2 <= h->d[0]) ex_search(m, far, q, h);
}
static int kheap_finish(kheap *h, int32_t idx[63], double d2[63]) {
    /* ascending (d2, idx): simple insertion sort */
    int n = h->n;
    for (int a = 1; a < n; a++) {
        double d = h->d[a]; int32_t i = h->i[a]; int b = a - 1;
        while (b >= 0 && key_less(d, i, h->d[b], h->i[b])) { h->d[b + 1] = h->d[b]; h->i[b + 1] = h->i[b]; b--; }
        h->d[b + 1] = d; h->i[b + 1] = i;
    }
    for (int a = 0; a < 63; a++) { idx[a] = a < n ? h->i[a] : -1; d2[a] = a < n ? h->d[a] : 0; }
    return n;
}
int orc_knn_exact(const orc_map *m, const double q[3], int32_t idx[63], double d2[63]) {
    kheap h; h.n = 0;
    if (m->n > 0) ex_search(m, 0, q, &h);
    return kheap_finish(&h, idx, d2);
}
int orc_knn_brute(const orc_map *m, const double q[3], int32_t idx[63], double d2[63]) {
    kheap h; h.n = 0;
    for (int i = 0; i < m->n; i++) kheap_offer(&h, photon_d2(&m->ph[i], q), i);
    return kheap_finish(&h, idx, d2);
}

/* One photon's term of the sums at src/Raytrace.cpp:100-125 (global) / :139-164 (caustic). */
static void photon_term(const orc_photon *ph, int map_kind, const double p[3], const double nrm[3],
                        double pd2, double r, double out[3]) {
    out[0] = out[1] = out[2] = 0.0;
    double d = -(nrm[0] * ph->dx + nrm[1] * ph->dy + nrm[2] * ph->dz);
    if (d <= 0) return;
    double dist[3] = {p[0] - ph->x, p[1] - ph->y, p[2] - ph->z};
    double k = (dist[0] * nrm[0] + dist[1] * nrm[1] + dist[2] * nrm[2]) /
               (dist[2] * dist[2] + dist[2] * dist[2] + dist[2] * dist[2]);
    dist[0] *= k; dist[1] *= k; dist[2] *= k;
    if (dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2] > 0.1) return;
    double filter = (1 - pd2 / r);
    if (map_kind == ORC_MAP_GLOBAL) {
        filter = filter * filter;
        out[0] = d * (ph->power[0] / 255.0) * filter;
        out[1] = d * (ph->power[1] / 255.0) * filter;
        out[2] = d * (ph->power[2] / 255.0) * filter;
    } else {
        filter = filter * filter * filter * filter;
        /* min() here is raytrace::min(int32,int32): the double is truncated */
        out[0] = d * filter * imin((int32_t)(ph->power[0] / 255.0f + 0.5), 1);
        out[1] = d * filter * imin((int32_t)(ph->power[1] / 255.0f + 0.5), 1);
        out[2] = d * filter * imin((int32_t)(ph->power[2] / 255.0f + 0.5), 1);
    }
}

void orc_gather(const orc_map *m, int map_kind, int order, const double p[3], const double nrm[3],
                double rgb[3]) {
    int32_t idx[63]; double d2[63];
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    double norm_k = map_kind == ORC_MAP_GLOBAL ? (3.141592653589 * 2) : (3.141592653589 * 8);
    if (order == ORC_ORDER_REFERENCE) {
        int found = orc_knn_ref(m, p, idx, d2);
        if (idx[0] < 0) return;
        double r = d2[0];
        for (int i = 0; i < found; i++) {
            if (idx[i] < 0) break;
            double t[3];
            photon_term(&m->ph[idx[i]], map_kind, p, nrm, d2[i], r, t);
            /* "x += 0" for skipped photons is exact, so adding the zero term is the same as `continue` */
            rgb[0] += t[0]; rgb[1] += t[1]; rgb[2] += t[2];
        }
        rgb[0] /= norm_k * r; rgb[1] /= norm_k * r; rgb[2] /= norm_k * r;
    } else {
        /* canonical (order-independent) sum: r = largest of the selected d2 (the reference's
         * distances[0]); every term is truncated to a multiple of 2^-56 and the terms are added as
         * 64-bit integers (0 <= term <= 1, at most 63 terms: no overflow), so the result does not
         * depend on the order in which an index delivers the photons.  Photons at exactly d2 == r have
         * weight (1 - d2/r) == 0, so ties at the k-th distance cannot change the sum either. */
        int found = orc_knn_exact(m, p, idx, d2);
        if (found == 0) return;
        double r = d2[found - 1];
        int64_t acc[3] = {0, 0, 0};
        for (int i = 0; i < found; i++) {
            if (!(d2[i] < r)) continue;
            double t[3];
            photon_term(&m->ph[idx[i]], map_kind, p, nrm, d2[i], r, t);
            for (int ch = 0; ch < 3; ch++) acc[ch] += (int64_t)(t[ch] * 72057594037927936.0);
        }
        for (int ch = 0; ch < 3; ch++) rgb[ch] = ((double)acc[ch] * (1.0 / 72057594037927936.0)) / (norm_k * r);
    }
}

typedef struct gb_job { const orc_map *m; int kind, order, lo, hi; const double *p, *nrm; double *rgb; } gb_job;
static void *gb_worker(void *v) {
    gb_job *j = (gb_job *)v;
    for (int i = j->lo; i < j->hi; i++) orc_gather(j->m, j->kind, j->order, j->p + 3 * i, j->nrm + 3 * i, j->rgb + 3 * i);
    return NULL;
}
void orc_gather_batch(const orc_map *m, int map_kind, int order, int count, const double *p,
                      const double *nrm, double *rgb, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; gb_job jobs[256];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (gb_job){m, map_kind, order, (int)((long long)count * t / nthreads),
                           (int)((long long)count * (t + 1) / nthreads), p, nrm, rgb};
        pthread_create(&th[t], NULL, gb_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* ----------------------------------------------------------------- trace */
typedef struct trace_ctx {
    const orc_prim *prims; int n;
    const orc_map *gmap, *cmap;
    const orc_render_params *rp;
    uint64_t K; uint64_t pixel;
    orc_tape *tape; orc_stats *stats;
} trace_ctx;

/* absorb branch, src/Raytrace.cpp:84-243 */
static uint32_t shade_absorb(trace_ctx *c, uint64_t IK, const double p[3], const double nrm[3], int obj) {
    const orc_prim *o = &c->prims[obj];
    double gi[3] = {0, 0, 0}, ci[3] = {0, 0, 0};
    if (c->stats) c->stats->absorb_points++;
    if (!(c->rp->flags & ORC_FLAG_NO_GLOBAL) && c->gmap) {
        orc_gather(c->gmap, ORC_MAP_GLOBAL, c->rp->order, p, nrm, gi);
        if (c->stats) c->stats->gathers++;
    }
    if (!(c->rp->flags & ORC_FLAG_NO_CAUSTIC) && c->cmap) {
        orc_gather(c->cmap, ORC_MAP_CAUSTIC, c->rp->order, p, nrm, ci);
        if (c->stats) c->stats->gathers++;
    }
    int light_count = 0;
    double shadow_ray[3] = {0, 0, 0}, result[3] = {0, 0, 0}, normal[3] = {0, 0, 0};
    for (int i = 0; i < 25; i++) {
        double x0 = draw(IK, 1 + 3 * i, c->tape) * 2 - 1;
        double z0 = draw(IK, 2 + 3 * i, c->tape) * 2 + 3;
        double y0 = 4.95;
        shadow_ray[0] = x0 - p[0]; shadow_ray[1] = y0 - p[1]; shadow_ray[2] = z0 - p[2];
        double max_dist = shadow_ray[0] * shadow_ray[0] + shadow_ray[1] * shadow_ray[1] + shadow_ray[2] * shadow_ray[2];
        normalize3(shadow_ray);
        double dt = draw(IK, 3 + 3 * i, c->tape);
        if (c->stats) c->stats->shadow_queries++;
        for (int k = 0; k < c->n; k++) {
            if (k == obj) continue;
            if (orc_prim_intersect(&c->prims[k], p, shadow_ray, dt, result, normal)) {
                double ds = (p[0] - result[0]) * (p[0] - result[0]) + (p[1] - result[1]) * (p[1] - result[1]) +
                            (p[2] - result[2]) * (p[2] - result[2]);
                if (ds > max_dist) continue;
                light_count++;
                break;
            }
        }
    }
    double direct = 0;
    if (light_count < 25) {
        double d = nrm[0] * shadow_ray[0] + nrm[1] * shadow_ray[1] + nrm[2] * shadow_ray[2];
        if (d > 0) direct += d * 0.2 * (1 - (light_count / (double)25));
        /* the Blinn term at :215-227 is computed into `specular`, which is never read: dead code */
    }
    if (c->rp->flags & ORC_FLAG_NO_DIRECT) direct = 0;
    double rr = direct + ci[0] + gi[0]; if (rr > 1) rr = 1;
    double gr = direct + ci[1] + gi[1]; if (gr > 1) gr = 1;
    double br = direct + ci[2] + gi[2]; if (br > 1) br = 1;
    uint32_t red = (uint32_t)orc_fastfloor(o->red * rr * 0xFF);
    uint32_t green = (uint32_t)orc_fastfloor(o->green * gr * 0xFF);
    uint32_t blue = (uint32_t)orc_fastfloor(o->blue * br * 0xFF);
    return (0xFFu << 24) | (red << 16) | (green << 8) | blue;
}

/* src/Raytrace.cpp:32-260 ; src/ray are shared, mutable, across the recursion */
static uint32_t trace_ray(trace_ctx *c, double src[3], double ray[3], int exclude, int bounce, uint32_t node) {
    if (bounce > 3) return 0xFF000000u;
    uint64_t IK = orc_item_key(c->K, c->pixel * 32u + node);
    double res[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
    double dt = draw(IK, 0, c->tape);
    if (c->stats) c->stats->closest_hit_queries++;
    int obj = orc_scene_intersect(c->prims, c->n, src, ray, exclude, dt, res, nrm);
    if (obj < 0) return 0xFF000000u;
    if (res[1] > 4.95 && res[0] > -1 && res[0] < 1 && res[2] > 3 && res[2] < 5) {
        const double *lc = c->rp->light_color;
        return (0xFFu << 24) | ((uint32_t)imin(orc_fastfloor(lc[0] * 0xFF) + 50, 0xFF) << 16) |
               ((uint32_t)imin(orc_fastfloor(lc[1] * 0xFF) + 50, 0xFF) << 8) |
               (uint32_t)imin(orc_fastfloor(lc[2] * 0xFF) + 50, 0xFF);
    }
    const orc_prim *o = &c->prims[obj];
    uint32_t refract_res = 0, reflect_res = 0, absorb_res = 0;
    if (o->transmission_chance > 0) {
        double n = 1 / o->refraction;
        double d = nrm[0] * ray[0] + nrm[1] * ray[1] + nrm[2] * ray[2];
        double n1[3] = {nrm[0] * d, nrm[1] * d, nrm[2] * d};
        n1[0] = ray[0] - n1[0]; n1[1] = ray[1] - n1[1]; n1[2] = ray[2] - n1[2];
        n1[0] *= n; n1[1] *= n; n1[2] *= n;
        double s = 1 - (n * n) * (1 - d * d);
        double sq = sqrt(s);
        double n2[3] = {nrm[0] * sq, nrm[1] * sq, nrm[2] * sq};
        n1[0] += -n2[0]; n1[1] += -n2[1]; n1[2] += -n2[2];
        src[0] = res[0] + n1[0] * 0.01; src[1] = res[1] + n1[1] * 0.01; src[2] = res[2] + n1[2] * 0.01;
        ray[0] = n1[0]; ray[1] = n1[1]; ray[2] = n1[2];
        normalize3(ray);
        refract_res = trace_ray(c, src, ray, -1, bounce + 1, node * 2);
    }
    if (o->specular_chance > 0) {
        double n1[3] = {nrm[0], nrm[1], nrm[2]};
        double k = n1[0] * ray[0] + n1[1] * ray[1] + n1[2] * ray[2];
        n1[0] *= k; n1[1] *= k; n1[2] *= k;
        n1[0] *= 2; n1[1] *= 2; n1[2] *= 2;
        src[0] = res[0]; src[1] = res[1]; src[2] = res[2];
        ray[0] = ray[0] - n1[0]; ray[1] = ray[1] - n1[1]; ray[2] = ray[2] - n1[2];
        normalize3(ray);
        reflect_res = trace_ray(c, src, ray, obj, bounce + 1, node * 2 + 1);
    }
    if (o->absorb_chance > 0) absorb_res = shade_absorb(c, IK, res, nrm, obj);
    float sred = ((reflect_res >> 16) & 0xFF) / 255.0f;
    float sgreen = ((reflect_res >> 8) & 0xFF) / 255.0f;
    float sblue = ((reflect_res) & 0xFF) / 255.0f;
    float tred = ((refract_res >> 16) & 0xFF) / 255.0f;
    float tgreen = ((refract_res >> 8) & 0xFF) / 255.0f;
    float tblue = ((refract_res) & 0xFF) / 255.0f;
    float ared = ((absorb_res >> 16) & 0xFF) / 255.0f;
    float agreen = ((absorb_res >> 8) & 0xFF) / 255.0f;
    float ablue = ((absorb_res) & 0xFF) / 255.0f;
    uint32_t red = (uint32_t)orc_fastfloor((sred * o->specular_chance + tred * o->transmission_chance + ared * o->absorb_chance) * 255);
    uint32_t green = (uint32_t)orc_fastfloor((sgreen * o->specular_chance + tgreen * o->transmission_chance + agreen * o->absorb_chance) * 255);
    uint32_t blue = (uint32_t)orc_fastfloor((sblue * o->specular_chance + tblue * o->transmission_chance + ablue * o->absorb_chance) * 255);
    return (0xFFu << 24) | (red << 16) | (green << 8) | blue;
}

/* loop body of render_task, src/Raytrace.cpp:285-297 */
uint32_t orc_trace_pixel(const orc_prim *prims, int n, const orc_map *gmap, const orc_map *cmap,
                         const orc_render_params *rp, int x, int y, orc_tape *tape, orc_stats *stats) {
    trace_ctx c = {prims, n, gmap, cmap, rp, orc_stream_key(rp->seed, ORC_KIND_PIXEL),
                   (uint64_t)x + (uint64_t)y * (uint64_t)rp->width, tape, stats};
    uint64_t IK0 = orc_item_key(c.K, c.pixel * 32u);
    double src[3] = {rp->cam[0], rp->cam[1], rp->cam[2]};
    double fov = (rp->width / 1280.0) * 64.0;
    double x1 = draw(IK0, 0, tape) * 0.6 - 0.3;
    double y1 = draw(IK0, 1, tape) * 0.6 - 0.3;
    double x0 = (x - rp->width / 2 + x1) / fov - src[0];
    double y0 = (y - rp->height / 2 + y1) / fov - src[1];
    double ray[3] = {x0, y0, -src[2]};
    normalize3(ray);
    return trace_ray(&c, src, ray, -1, 0, 1);
}

/* row-job queue restating src/JobSystem.cpp:37-73 (workers take up to 10 jobs per lock) */
typedef struct row_pool {
    pthread_mutex_t mu; int next, y0, y1, ystep;
    const orc_prim *prims; int n; const orc_map *gmap, *cmap; const orc_render_params *rp;
    uint32_t *pane; orc_stats stats;
} row_pool;
static void *row_worker(void *v) {
    row_pool *rp = (row_pool *)v;
    orc_stats local; memset(&local, 0, sizeof(local));
    for (;;) {
        int first, count = 0;
        pthread_mutex_lock(&rp->mu);
        first = rp->next;
        while (count < 10 && rp->next < rp->y1) { rp->next += rp->ystep; count++; }
        pthread_mutex_unlock(&rp->mu);
        if (count == 0) break;
        for (int j = 0; j < count; j++) {
            int y = first + j * rp->ystep;
            for (int x = 0; x < rp->rp->width; x++)
                rp->pane[(size_t)x + (size_t)y * rp->rp->width] =
                    orc_trace_pixel(rp->prims, rp->n, rp->gmap, rp->cmap, rp->rp, x, y, NULL, &local);
        }
    }
    pthread_mutex_lock(&rp->mu);
    rp->stats.closest_hit_queries += local.closest_hit_queries;
    rp->stats.shadow_queries += local.shadow_queries;
    rp->stats.gathers += local.gathers;
    rp->stats.absorb_points += local.absorb_points;
    pthread_mutex_unlock(&rp->mu);
    return NULL;
}
void orc_render_rows(const orc_prim *prims, int n, const orc_map *gmap, const orc_map *cmap,
                     const orc_render_params *rp, int y0, int y1, int ystep, int nthreads,
                     uint32_t *pane, orc_stats *stats) {
    row_pool pool;
    memset(&pool, 0, sizeof(pool));
    pthread_mutex_init(&pool.mu, NULL);
    if (ystep < 1) ystep = 1;
    pool.next = y0; pool.y0 = y0; pool.y1 = y1; pool.ystep = ystep;
    pool.prims = prims; pool.n = n; pool.gmap = gmap; pool.cmap = cmap; pool.rp = rp; pool.pane = pane;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, row_worker, &pool);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&pool.mu);
    if (stats) {
        stats->closest_hit_queries += pool.stats.closest_hit_queries;
        stats->shadow_queries += pool.stats.shadow_queries;
        stats->gathers += pool.stats.gathers;
        stats->absorb_points += pool.stats.absorb_points;
    }
}
