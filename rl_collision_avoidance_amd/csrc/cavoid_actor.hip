// cavoid_actor.hip -- C ABI (include/cavoid.h, cavoid_actor_run) over actor_kernel<N> (cavoid_actor.hpp): K closed-loop GA3C actor
// steps -- policy forward, action selection, env.step, experience bookkeeping -- in ONE launch.  Own translation unit (the
// kernel carries the policy's GEMM loops and the env step; instantiated per agent count).
#include <cstdlib>
#define CAVOID_ACTOR_KERNELS
#include "cavoid_actor_host.hpp"
#include "cavoid_policy_host.hpp"
#include "cavoid_rollout_host.hpp"

using namespace cavoid;

static int actor_run(cavoid_env *e, cavoid_policy *h, cavoid_policy *frozen, cavoid_rollout *r, const cavoid_rollout_buffers *b, float *obs_cur, float *obs_next,
                     float *rewards, uint8_t *done, uint8_t *game_over, int32_t *actions, float *values, int32_t n_steps,
                     int32_t greedy, void *stream) {
    if (!e || !h || !r || !b || b->struct_size != (int32_t)sizeof(cavoid_rollout_buffers) || !obs_cur || !obs_next || obs_cur == obs_next ||
        !rewards || !done || !game_over || !actions || !values || n_steps < 0)
        return CAVOID_EINVAL;
    if (!b->x || !b->val || !b->ret || !b->act || !b->emit_t || !b->dup_x || !b->dup_r || !b->dup_a || !b->dup_src || !b->dup_count ||
        !b->ep_out || !b->ep_count || b->dup_capacity < 1 || b->ep_capacity < 1)
        return CAVOID_EINVAL;
    if (n_steps == 0) return CAVOID_OK;
    if (!h->loaded) return CAVOID_EINVAL;
    // the three handles must describe the same batch
    if (h->device != e->device || r->device != e->device || r->c.num_slots != e->A || r->c.max_agents != e->cfg.max_agents ||
        r->c.obs_width != e->k.width || h->in_size != e->k.width - 1 || h->max_other != e->cfg.max_other)
        return CAVOID_EINVAL;
    // what the fused kernel does not carry (the step-by-step entry points do): velocity actions, the float32-MFMA inference kernel
    // or a non-default number of split products (the kernel carries the default form of cavoid_policy_forward, so that both stay
    // bit-identical); frozen-network agents need a second network (BatchedRollout keeps those on the step-by-step path)
    if (e->cfg.dynamics == CAVOID_DYN_HOLONOMIC || !h->use_split || h->split_products != kSpDefaultProducts) return CAVOID_EUNSUPPORTED;
    // frozen-network agents act by THEIR network: without it this entry point would hand them the learner's sample
    if (!frozen && e->cfg.gen_frozen_fraction > 0.0 && e->cfg.gen_nonlearning_fraction > 0.0) return CAVOID_EUNSUPPORTED;
    if (frozen && (!frozen->loaded || frozen->device != e->device || frozen->in_size != h->in_size || frozen->max_other != h->max_other ||
                   frozen->num_actions != h->num_actions || !frozen->use_split || frozen->split_products != kSpDefaultProducts))
        return frozen->loaded ? CAVOID_EUNSUPPORTED : CAVOID_EINVAL;
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_a = cavoid_ahead_prepare(e, n_steps, static_cast<hipStream_t>(stream))) return rc_a;     // (scenario look-ahead: the rings cover n_steps restarts)
    cavoid_ahead_consumed(e, n_steps);
    const KCfg &k = e->k;
    // ORCA agents / box scenarios generated inside the step: the env step's RVO instantiation (as cavoid_step_autoreset routes them)
    const bool rvo_form = e->cfg.rvo_enabled || (e->cfg.gen_mode == 1 && e->pool_size <= 0);
    int tile = (k.tile_rows * k.width + 3) & ~3;
    if (tile < k.park_floats) tile = k.park_floats;
    // the env step borrows the (idle) activation planes: staging arrays + obs tile (+ the ORCA lines, 64 (N-1) x 16 floats: N <= 12)
    if (actor_env_lds_bytes(e->cfg.max_agents, tile, k.rvo_lds_floats) > (size_t)2 * kSpPlaneB) return CAVOID_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);

    PolicyArgs a{};
    a.rows = e->A; a.stride = k.width; a.max_other = h->max_other; a.num_actions = h->num_actions; a.in_size = h->in_size;
    a.avg = h->normalize ? h->avg : nullptr; a.std = h->normalize ? h->std : nullptr;
    a.frags = h->frags; a.bias = h->bias; a.min_policy = h->min_policy;
    a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32);
    a.step_counter = h->step_counter; a.blocks_done = h->blocks_done; a.cu_tickets = h->cu_tickets;
    const SplitArgs sa{a, h->sfrags, h->sbias};
    SplitArgs fz = sa;                                       // the frozen network: same shapes, its own weights / normalisation; argmax only
    if (frozen) {
        fz.p.avg = frozen->normalize ? frozen->avg : nullptr; fz.p.std = frozen->normalize ? frozen->std : nullptr;
        fz.p.frags = frozen->frags; fz.p.bias = frozen->bias; fz.p.min_policy = frozen->min_policy;
        fz.sfrags = frozen->sfrags; fz.sbias = frozen->sbias;
    }
    RolloutCfg rc = r->c;
    rc.dup_capacity = b->dup_capacity; rc.ep_capacity = b->ep_capacity;
    RolloutIO rio{};
    rio.step = -1; rio.x = b->x; rio.val = b->val; rio.ret = b->ret; rio.act = b->act; rio.emit_t = b->emit_t;
    rio.dup_x = b->dup_x; rio.dup_r = b->dup_r; rio.dup_a = b->dup_a; rio.dup_src = b->dup_src; rio.dup_count = b->dup_count;
    rio.ep_out = b->ep_out; rio.ep_count = b->ep_count;
    ActorIO io{};
    io.obs[0] = obs_cur; io.obs[1] = obs_next; io.rewards = rewards; io.done = done; io.game_over = game_over;
    io.actions = actions; io.values = values; io.rollout_step = r->s.step_counter; io.n_steps = n_steps; io.greedy = greedy ? 1 : 0;
    // the tile's env step spread over the workgroup's four wavefronts (cavoid_quad.hpp) where that form carries the configuration: not the
    // ORCA / in-step box instantiation, one pass per tile, its LDS inside the lent planes (CAVOID_ACTOR_QUAD=0: wavefront 0 alone, A/B runs)
    io.quad = (!rvo_form && e->cfg.max_agents <= kActorQuadMaxAgents && k.tile_rows >= k.wpw * e->cfg.max_agents &&
               quad_lds_bytes<kActorQuadMaxAgents>((k.tile_rows * k.width + 3) & ~3) <= (size_t)2 * kSpPlaneB) ? 1 : 0;
    if (const char *ov = std::getenv("CAVOID_ACTOR_QUAD")) io.quad = (io.quad && std::atoi(ov) != 0) ? 1 : 0;
    const int rc_launch = frozen ? cavoid_launch_actor_frozen(e, sa, fz, rc, r->s, rio, io, s)
                                 : (rvo_form ? cavoid_launch_actor_rvo(e, sa, rc, r->s, rio, io, s) : launch_actor_any<false>(e, sa, sa, rc, r->s, rio, io, s));
    if (rc_launch != CAVOID_OK) return rc_launch;
    hipLaunchKernelGGL(actor_finish_kernel, dim3(1), dim3(1), 0, s, r->s.step_counter, h->step_counter, n_steps);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

extern "C" int cavoid_actor_run(cavoid_env *e, cavoid_policy *h, cavoid_rollout *r, const cavoid_rollout_buffers *b, float *obs_cur, float *obs_next,
                                float *rewards, uint8_t *done, uint8_t *game_over, int32_t *actions, float *values, int32_t n_steps,
                                int32_t greedy, void *stream) {
    return actor_run(e, h, nullptr, r, b, obs_cur, obs_next, rewards, done, game_over, actions, values, n_steps, greedy, stream);
}

extern "C" int cavoid_actor_run_mix(cavoid_env *e, cavoid_policy *h, cavoid_policy *frozen, cavoid_rollout *r, const cavoid_rollout_buffers *b,
                                    float *obs_cur, float *obs_next, float *rewards, uint8_t *done, uint8_t *game_over, int32_t *actions,
                                    float *values, int32_t n_steps, int32_t greedy, void *stream) {
    if (!frozen) return CAVOID_EINVAL;
    return actor_run(e, h, frozen, r, b, obs_cur, obs_next, rewards, done, game_over, actions, values, n_steps, greedy, stream);
}

extern "C" int cavoid_step_push(cavoid_env *e, cavoid_rollout *r, const cavoid_rollout_buffers *b, const float *obs_cur, float *obs_next,
                                const int32_t *actions, const float *values, float *rewards, uint8_t *done, uint8_t *game_over, int32_t step,
                                void *stream) {
    if (!e || !r || !b || b->struct_size != (int32_t)sizeof(cavoid_rollout_buffers) || !obs_cur || !obs_next || obs_cur == obs_next ||
        !actions || !values || !rewards || !done || !game_over)
        return CAVOID_EINVAL;
    if (!b->x || !b->val || !b->ret || !b->act || !b->emit_t || !b->dup_x || !b->dup_r || !b->dup_a || !b->dup_src || !b->dup_count ||
        !b->ep_out || !b->ep_count || b->dup_capacity < 1 || b->ep_capacity < 1)
        return CAVOID_EINVAL;
    if (r->device != e->device || r->c.num_slots != e->A || r->c.max_agents != e->cfg.max_agents || r->c.obs_width != e->k.width) return CAVOID_EINVAL;
    if (e->cfg.dynamics == CAVOID_DYN_HOLONOMIC) return CAVOID_EUNSUPPORTED;       // (velocity actions: cavoid_step_continuous + cavoid_rollout_push)
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc_a = cavoid_ahead_prepare(e, 1, s)) return rc_a;
    cavoid_ahead_consumed(e, 1);
    RolloutCfg rc = r->c;
    rc.dup_capacity = b->dup_capacity; rc.ep_capacity = b->ep_capacity;
    RolloutIO rio{};
    rio.step = step; rio.x = b->x; rio.val = b->val; rio.ret = b->ret; rio.act = b->act; rio.emit_t = b->emit_t;
    rio.dup_x = b->dup_x; rio.dup_r = b->dup_r; rio.dup_a = b->dup_a; rio.dup_src = b->dup_src; rio.dup_count = b->dup_count;
    rio.ep_out = b->ep_out; rio.ep_count = b->ep_count;
    ActorIO io{};
    io.obs[0] = const_cast<float *>(obs_cur); io.obs[1] = obs_next; io.rewards = rewards; io.done = done; io.game_over = game_over;
    io.actions = const_cast<int32_t *>(actions); io.values = const_cast<float *>(values); io.rollout_step = r->s.step_counter; io.n_steps = 1;
    const bool rvo_form = e->cfg.rvo_enabled || (e->cfg.gen_mode == 1 && e->pool_size <= 0);
    const int rc_launch = rvo_form ? cavoid_launch_step_push_rvo(e, rc, r->s, rio, io, step, s) : launch_step_push_any<false>(e, rc, r->s, rio, io, step, s);
    if (rc_launch != CAVOID_OK) return rc_launch;
    if (step < 0) {                                          // the device-side step counter moves on (hipGraph replays)
        hipLaunchKernelGGL(actor_finish_kernel, dim3(1), dim3(1), 0, s, r->s.step_counter, static_cast<int32_t *>(nullptr), 1);
        HIP_TRY(hipGetLastError());
    }
    return CAVOID_OK;
}

#ifdef CAVOID_TRACE
// development build only: the phase stamps of actor_kernel<N, false, false> (this translation unit's copy of g_pol_trace; tools/trace_actor.py)
extern "C" int cavoid_actor_debug_trace(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(cavoid::g_pol_trace), &dev_ptr, sizeof(dev_ptr)));
    return CAVOID_OK;
}
#endif
