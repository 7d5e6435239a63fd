// The -DCAVOID_DEV_PKFORM=<n> variants of neighbour_features() (csrc/cavoid_kernels.hpp) that round 5's bisect of the packed-float32
// failure was run with (tools/experiments/pk_opsel_bisect.sh; results: profiles/r05_b_pk_bisect.txt).  Only form 0 -- the round-4 source
// left to the vectoriser, whose code object tools/experiments/pk_isa_patch.py edits at the ISA level -- is still in the product
// source; the others are kept here for the record (drop this block in place of the `#elif CAVOID_DEV_PKFORM == 0` branch to rebuild them).
#if !defined(CAVOID_DEV_PKFORM)
    float t_par = q.vyf * pyf, t_orth = q.vxf * pyf;
    asm volatile("" : "+v"(t_par), "+v"(t_orth));
    f[2] = __builtin_fmaf(q.vxf, pxf, t_par);
    f[3] = __builtin_fmaf(q.vyf, pxf, -t_orth);
#elif CAVOID_DEV_PKFORM == 0
    // development (tools/experiments/pk_opsel_bisect.sh): the round-4 source, left to the vectoriser
    f[2] = __builtin_fmaf(q.vxf, pxf, q.vyf * pyf);
    f[3] = __builtin_fmaf(q.vyf, pxf, -(q.vxf * pyf));
#else
    // development: the packed pair spelled out -- (vx, vy) * (py, py), then (vx, vy) * (px, px) + swap(product) with the high half
    // negated -- with what the bisect varies between and around the two instructions
    typedef float pkf2 __attribute__((ext_vector_type(2)));
    pkf2 v2 = {q.vxf, q.vyf}, py2 = {pyf, pyf}, px2 = {pxf, pxf}, t2, r2;
#define CAVOID_PK_STR2(x) #x
#define CAVOID_PK_STR(x) CAVOID_PK_STR2(x)
#if CAVOID_DEV_PKFORM == 1          /* swapped-halves fma, CAVOID_DEV_PKNOPS wait states between the two (-1: none) */
    asm volatile(
#if defined(CAVOID_DEV_PKDRAIN)
        "s_waitcnt lgkmcnt(0)\n\t"
#endif
        "v_pk_mul_f32 %1, %2, %3\n\t"
#if CAVOID_DEV_PKNOPS >= 0
        "s_nop " CAVOID_PK_STR(CAVOID_DEV_PKNOPS) "\n\t"
#endif
        "v_pk_fma_f32 %0, %2, %4, %1 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]"
        : "=&v"(r2), "=&v"(t2) : "v"(v2), "v"(py2), "v"(px2));
#elif CAVOID_DEV_PKFORM == 2        /* the same arithmetic, the swap made by two moves: no op_sel on the packed fma */
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t2) : "v"(v2), "v"(py2));
    pkf2 s2 = {t2.y, t2.x};
    asm volatile("" : "+v"(s2));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_hi:[0,0,1]" : "=v"(r2) : "v"(v2), "v"(px2), "v"(s2));
#elif CAVOID_DEV_PKFORM == 3        /* the swapped-halves pair as TWO asm statements: the scheduler is free to put work between them */
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t2) : "v"(v2), "v"(py2));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]" : "=v"(r2) : "v"(v2), "v"(px2), "v"(t2));
#elif CAVOID_DEV_PKFORM == 4        /* two statements, the swap moved into the MUL's source (old registers): the fma reads the fresh pair straight */
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t2) : "v"(v2), "v"(py2));   // (vy*py, -vx*py)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r2) : "v"(v2), "v"(px2), "v"(t2));
#elif CAVOID_DEV_PKFORM == 5        /* the pair, then an LDS read INTO the fma's third source pair right behind it (write after read) */
    {
        int zero = 0;
        asm volatile(
            "v_pk_mul_f32 %1, %2, %3\n\t"
            "v_pk_fma_f32 %0, %2, %4, %1 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]\n\t"
#if defined(CAVOID_DEV_PKGAP)
            "s_nop 7\n\ts_nop 7\n\t"
#endif
            "ds_read_b64 %1, %5\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(r2), "=&v"(t2) : "v"(v2), "v"(py2), "v"(px2), "v"(zero) : "memory");
    }
#elif CAVOID_DEV_PKFORM == 6        /* as 5, the swap in the mul's old sources: a straight fma, then the LDS read into its third source */
    {
        int zero = 0;
        asm volatile(
            "v_pk_mul_f32 %1, %2, %3 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %2, %4, %1\n\t"
            "ds_read_b64 %1, %5\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(r2), "=&v"(t2) : "v"(v2), "v"(py2), "v"(px2), "v"(zero) : "memory");
    }
#elif CAVOID_DEV_PKFORM == 7        /* the swapped-halves fma IN PLACE: its destination pair is its third source pair (what the compiler
                                       made of neighbour slots 1 and 2 -- v_pk_fma_f32 v[34:35], v[38:39], v[82:83], v[34:35] op_sel:[0,0,1]) */
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t2) : "v"(v2), "v"(py2));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]" : "+v"(t2) : "v"(v2), "v"(px2));
    r2 = t2;
#elif CAVOID_DEV_PKFORM == 8        /* in place WITHOUT the swap (the swap made in the mul's old sources) */
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t2) : "v"(v2), "v"(py2));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t2) : "v"(v2), "v"(px2));
    r2 = t2;
#endif
    f[2] = r2.x;
    f[3] = r2.y;
#endif
