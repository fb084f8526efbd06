"""CPU oracle of the ImageRegistrator optimizer loop (TEST INFRASTRUCTURE ONLY).

Orchestrates the C per-observation oracle (oracle_reg.c) exactly as the reference's host code does:
  opt::Optimizer::RunOnCurrentScale                     src/opt/optimizer.cc:49-182
  IntrinsicsAndPoseOptimizer::Apply / CreateDeltaState /
    ComputeResidualForState / CountAndIndexVariables    src/opt/intrinsics_and_pose_optimizer.cc:48-259,385-558
  VisibilityEstimator::CreateObservationsForAllImages   src/opt/visibility_estimator.cc:49-91
  ColorOptimizer::Apply                                 src/opt/color_optimizer.cc:40-123
  CostCalculator::ComputeCost, Problem::ComputeCost     src/opt/cost_calculator.cc:44-100, src/opt/problem.cc:602-631
Only numpy-array plumbing happens in Python; every per-point / per-observation loop is in C.  Non-rig and rig images; PINHOLE / OPENCV / THIN_PRISM_FISHEYE.
Images are visited in ascending id (the reference's unordered_map order is unspecified).
"""
import numpy as np

from . import binding as ob
from . import reg_binding as rb

K_MANY = 100


class OracleRegProblem:
    def __init__(self, K=5, robust_type=1, robust_param=float(np.float32(30 * np.sqrt(5) / np.sqrt(2))), fixed_weight=1.0,
                 var_weight=1.0, max_valid_intensity=252.0, occlusion_threshold=0.01, splat_radius=0.03,
                 current_image_scale=0, image_scale_count=2, depth_weight=0.0, depth_robust_type=2, depth_robust_param=0.02):
        self.depth_weight = depth_weight; self.depth_robust_type = depth_robust_type; self.depth_robust_param = depth_robust_param
        self.K = K; self.robust_type = robust_type; self.robust_param = robust_param
        self.fixed_weight = fixed_weight; self.var_weight = var_weight
        self.max_valid_intensity = max_valid_intensity; self.occlusion_threshold = occlusion_threshold
        self.splat_radius = splat_radius
        self.current_image_scale = current_image_scale; self.image_scale_count = image_scale_count
        self.scales = {}; self.intr = {}; self.images = {}; self.splat = None
        self.rigs = {}         # rig_id -> list of (q, t) image_T_rig per camera (camera 0 = reference)
        self.frames = []       # RigImages: (rig_id, [image id per camera])
        self.obs = {}     # (image, scale) -> (idx, x, y, s, flags)

    # ---- state ------------------------------------------------------------------------------------------------------------
    def set_point_scale(self, s, xyz, radius, nbr, fixed_desc):
        n = len(xyz)
        self.scales[s] = dict(pts=np.ascontiguousarray(xyz, np.float32), radius=np.float32(radius), nbr=np.ascontiguousarray(nbr, np.uint32),
                              fixed=np.ascontiguousarray(fixed_desc, np.float32), var=np.zeros((n, self.K), np.float32),
                              counts=np.full(n, 99999, np.int32))

    def set_variable_descriptors(self, s, desc, counts):
        self.scales[s]["var"] = np.ascontiguousarray(desc, np.float32).copy()
        self.scales[s]["counts"] = np.ascontiguousarray(counts, np.int32).copy()

    def set_intrinsics(self, iid, w, h, params, min_image_scale, n_levels, model=0):
        self.intr[iid] = dict(w=w, h=h, params=np.ascontiguousarray(params, np.float32).copy(), min=min_image_scale, n=n_levels,
                              model=model)
        self._rebuild(iid)

    def _rebuild(self, iid):
        I = self.intr[iid]
        I["levels"] = rb.camera_pyramid(rb.make_camera(I["w"], I["h"], I["params"], I["model"]), I["n"])

    def set_image(self, image_id, iid, pyr, masks=None):
        self.images[image_id] = dict(intr=iid, pyr=pyr, masks=masks, q=np.array([1, 0, 0, 0], np.float32), t=np.zeros(3, np.float32))

    def set_depth_maps(self, image_id, levels):
        """inverse-free depth maps of the image, one per pyramid level of its camera (Image::depth_maps_, src/opt/image.h)"""
        self.images[image_id]["depth"] = [np.ascontiguousarray(d, np.float32) for d in levels]

    def set_image_pose(self, image_id, q, t):
        self.images[image_id]["q"] = np.ascontiguousarray(q, np.float32).copy()
        self.images[image_id]["t"] = np.ascontiguousarray(t, np.float32).copy()
        if self.frames:
            self._compose_rig_poses()

    def get_image_pose(self, image_id):
        return self.images[image_id]["q"].copy(), self.images[image_id]["t"].copy()

    # ---- rigs (src/opt/rig.h:41-76, problem.h RigImages) --------------------------------------------------------------------
    def set_rig(self, rig_id, image_T_rig):
        self.rigs[rig_id] = [(np.ascontiguousarray(q, np.float32).copy(), np.ascontiguousarray(t, np.float32).copy()) for q, t in image_T_rig]

    def add_rig_images(self, rig_id, image_ids):
        self.frames.append((rig_id, list(image_ids)))
        for c, iid in enumerate(image_ids):
            self.images[iid]["rig"] = (rig_id, c, image_ids[0])
        self._compose_rig_poses()

    def get_rig(self, rig_id, camera):
        q, t = self.rigs[rig_id][camera]
        return q.copy(), t.copy()

    def _compose_rig_poses(self):
        """image_T_global of non-reference rig images = image_T_rig[c] * reference image_T_global (CreateDeltaState :539-548)."""
        for rig_id, ids in self.frames:
            ref = self.images[ids[0]]
            for c, iid in enumerate(ids):
                if c == 0:
                    continue
                q, t = rb.se3_mul(*self.rigs[rig_id][c], ref["q"], ref["t"])
                self.images[iid]["q"], self.images[iid]["t"] = q, t

    def _rig_link(self, image_id):
        r = self.images[image_id].get("rig")
        if r is None or r[1] == 0:
            return None
        ref = self.images[r[2]]
        return rb.rig_link(self.rigs[r[0]][r[1]][0], ref["q"], ref["t"])

    def set_splat_points(self, xyz):
        self.splat = np.ascontiguousarray(xyz, np.float32)

    def _R(self, im):
        return ob.quat_to_R(im["q"])

    def get_state(self):
        return ({k: v["params"].copy() for k, v in self.intr.items()}, {k: (v["q"].copy(), v["t"].copy()) for k, v in self.images.items()},
                {k: [(q.copy(), t.copy()) for q, t in v] for k, v in self.rigs.items()})

    def set_state(self, st):
        for k, p in st[0].items():
            self.intr[k]["params"] = p.copy(); self._rebuild(k)
        for k, (q, t) in st[1].items():
            self.images[k]["q"] = q.copy(); self.images[k]["t"] = t.copy()
        if len(st) > 2:
            self.rigs = {k: [(q.copy(), t.copy()) for q, t in v] for k, v in st[2].items()}

    # ---- steps -------------------------------------------------------------------------------------------------------------
    def _best_scale(self, I):
        return min(I["min"] + I["n"] - 1, max(I["min"], max(0, self.current_image_scale)))

    def _observe(self, image_id, s, image_scale, border, indices=None, depth=None):
        im = self.images[image_id]; I = self.intr[im["intr"]]; S = self.scales[s]
        o = rb.observe(S["pts"], float(S["radius"]), self._R(im), im["t"], I["levels"], I["min"], im["pyr"], im["masks"], depth, image_scale,
                       border, self.current_image_scale, self.image_scale_count, self.occlusion_threshold, self.max_valid_intensity,
                       indices=indices)
        f = rb.neighbors_observed(len(S["pts"]), o[0], S["nbr"], self.K)
        return o + (f,)

    # ---- ObservationsCache (src/opt/observations_cache.cc) -----------------------------------------------------------------
    def determine_observed_indices(self):
        """DetermineAndSaveObservedPointIndices (:104-125): full visibility at image scale 0 -> per image, per point scale lists."""
        old, old_cache = self.current_image_scale, getattr(self, "cache_observations", False)
        self.current_image_scale = 0; self.cache_observations = False
        self.update_observations(1)
        self.current_image_scale = old; self.cache_observations = old_cache
        self.observed = {}
        for image_id in sorted(self.images):
            self.observed[image_id] = {s: (np.asarray(self.obs[(image_id, s)][0], np.uint64) if (image_id, s) in self.obs
                                           else np.zeros(0, np.uint64)) for s in sorted(self.scales)}

    def update_observations(self, border=1):
        self.obs = {}
        cached = getattr(self, "cache_observations", False)
        for image_id in sorted(self.images):
            im = self.images[image_id]; I = self.intr[im["intr"]]
            scale = self._best_scale(I)
            depth = None if cached else rb.splat_depth(self.splat, self._R(im), im["t"], I["levels"][max(0, scale - I["min"])], self.splat_radius)
            had_many = False
            for s in sorted(self.scales, reverse=True):
                if cached:     # GetObservations (:52-68) -> AppendObservationsForIndexedPointsVisibleInImage
                    o = self._observe(image_id, s, scale, border, indices=np.asarray(self.observed[image_id][s], np.uint32))
                else:
                    o = self._observe(image_id, s, scale, border, depth=depth)
                self.obs[(image_id, s)] = o
                if len(o[0]) > K_MANY:
                    had_many = True
                elif len(o[0]) == 0 and had_many:
                    break

    def color_update(self):
        for s in sorted(self.scales):
            S = self.scales[s]
            S["var"][:] = 0; S["counts"][:] = 0
            for image_id in sorted(self.images):
                if (image_id, s) not in self.obs:
                    continue
                im = self.images[image_id]; I = self.intr[im["intr"]]; o = self.obs[(image_id, s)]
                rb.color_accumulate(len(S["pts"]), S["nbr"], self.K, I["min"], im["pyr"], o[:4], o[4], S["var"], S["counts"])
            rb.color_finish(self.K, S["var"], S["counts"])

    def _cost_value(self, sums, counts):
        use_f, use_v, use_d = self.fixed_weight > 0, self.var_weight > 0, self.depth_weight > 0
        r = 0.0
        if use_f and counts[0] > 0:
            r += self.fixed_weight * sums[0] / counts[0]
        if use_v and counts[1] > 0:
            r += self.var_weight * sums[1] / counts[1]
        if use_d and counts[2] > 0:                       # problem.cc:616-620
            r += self.depth_weight * sums[2] / counts[2]
        if (not use_f and not use_v and not use_d) or (counts[0] == 0 and counts[1] == 0 and counts[2] == 0):
            r = float("inf")
        return r

    def _cost_of(self, obs):
        sums = np.zeros(3); counts = np.zeros(3, np.int64)
        for image_id in sorted(self.images):
            im = self.images[image_id]; I = self.intr[im["intr"]]
            for s in sorted(self.scales):
                if (image_id, s) not in obs:
                    continue
                S = self.scales[s]; o = obs[(image_id, s)]
                s2, c2 = rb.cost(len(S["pts"]), S["nbr"], self.K, S["fixed"], S["var"], S["counts"], I["min"], im["pyr"], o[:4], o[4],
                                 self.robust_type, self.robust_param, self.fixed_weight, self.var_weight)
                sums[:2] += s2; counts[:2] += c2
                if self.depth_weight > 0:                 # cost_calculator.cc:221-245
                    sd, cd = rb.depth_cost(S["pts"], I["min"], im["depth"], im["q"], im["t"], o[:4], self.depth_robust_type,
                                           self.depth_robust_param)
                    sums[2] += sd; counts[2] += cd
        return sums, counts

    def compute_cost(self):
        sums, counts = self._cost_of(self.obs)
        if not counts.any():
            return float("inf")
        return self._cost_value(sums, counts)

    def apply(self, lam, print_progress=False):
        intr_index, image_index = {}, {}
        V = 0
        for k in sorted(self.intr):
            intr_index[k] = V; V += len(self.intr[k]["params"])
        rig_index = {}
        for k in sorted(self.rigs):                       # rig extrinsics, reference camera excluded (:455-460)
            rig_index[k] = V; V += 6 * (len(self.rigs[k]) - 1)
        for k in sorted(self.images):                     # poses: non-rig images and rig reference images (:461-472)
            r = self.images[k].get("rig")
            if r is not None and r[1] > 0:
                continue
            image_index[k] = V; V += 6
        H = np.zeros((V, V)); b = np.zeros(V)
        sums = np.zeros(3); counts = np.zeros(3, np.int64)
        vis = {}
        for image_id in sorted(self.images):
            im = self.images[image_id]; I = self.intr[im["intr"]]
            NI = len(I["params"])
            ii = intr_index[im["intr"]]
            link = self._rig_link(image_id)
            if link is None:
                pi = image_index[image_id]
                g = list(range(ii, ii + NI)) + list(range(pi, pi + 6))
            else:
                rid, cam_i, ref_id = im["rig"]
                ri = rig_index[rid] + 6 * (cam_i - 1); pi = image_index[ref_id]
                g = list(range(ii, ii + NI)) + list(range(ri, ri + 6)) + list(range(pi, pi + 6))
            for s in sorted(self.scales):
                if (image_id, s) not in self.obs:
                    continue
                S = self.scales[s]; o = self.obs[(image_id, s)]
                vis[(image_id, s)] = o[0].copy()
                Hl, bl, s2, c2 = rb.accumulate(S["pts"], float(S["radius"]), S["nbr"], self.K, S["fixed"], S["var"], S["counts"], I["levels"][0],
                                               I["min"], im["pyr"], self._R(im), im["t"], o[:4], o[4], self.robust_type, self.robust_param,
                                               self.fixed_weight, self.var_weight, rig=link)
                sums[:2] += s2; counts[:2] += c2
                for r in range(len(g)):
                    for c in range(r, len(g)):
                        H[g[r], g[c]] += Hl[r, c]
                    b[g[r]] += bl[r]
                if self.depth_weight > 0:                 # intrinsics_and_pose_optimizer.cc:747-757, 1150-1296
                    assert link is None, "depth residuals of dependent rig images: LOG(FATAL) in the reference (:1199-1207)"
                    res, JI, JP = rb.depth_rows(S["pts"], float(S["radius"]), I["levels"][0], I["min"], im["depth"], self._R(im), im["t"],
                                                im["q"], o[:4])
                    Hd, bd, sd, cd = rb.depth_accumulate(res, JI, JP, self.depth_robust_type, self.depth_robust_param, self.depth_weight)
                    sums[2] += sd; counts[2] += cd
                    gd = list(range(ii, ii + NI)) + list(range(pi, pi + 6))
                    for r in range(len(gd)):
                        for c in range(r, len(gd)):
                            H[gd[r], gd[c]] += Hd[r, c]
                        b[gd[r]] += bd[r]
        initial = self._cost_value(sums, counts)
        if print_progress:
            print("    Initial residual: %g (#fixed residuals: %d, #variable residuals: %d)" % (initial, counts[0], counts[1]))
        old = self.get_state()
        lam = np.float32(lam)
        for lm in range(10):
            Hl = H.copy()
            Hl[np.diag_indices(V)] *= (1 + lam)
            x = ob.ldlt_solve_upper(Hl, b)
            params = {k: (p + (-1 * x[intr_index[k]:intr_index[k] + len(p)])).astype(np.float32) for k, p in old[0].items()}
            poses = {k: (ob.se3_update(x[image_index[k]:image_index[k] + 6], q, t) if k in image_index else (q, t))
                     for k, (q, t) in old[1].items()}
            rigs = {k: [v[0]] + [ob.se3_update(x[rig_index[k] + 6 * (c - 1):rig_index[k] + 6 * c], *v[c]) for c in range(1, len(v))]
                    for k, v in old[2].items()}                                              # Rig::Update (rig.cc:9-23)
            self.set_state((params, poses, rigs))
            self._compose_rig_poses()
            trial_obs = {}
            for image_id in sorted(self.images):
                I = self.intr[self.images[image_id]["intr"]]
                scale = self._best_scale(I)
                had_many = False
                for s in sorted(self.scales, reverse=True):
                    idx = vis.get((image_id, s), np.zeros(0, np.uint32))
                    o = self._observe(image_id, s, scale, 1, indices=idx)
                    trial_obs[(image_id, s)] = o
                    if len(o[0]) > K_MANY:
                        had_many = True
                    elif len(o[0]) == 0 and had_many:
                        break
            ts, tc = self._cost_of(trial_obs)
            new = self._cost_value(ts, tc)
            if new < initial or lm == 9:
                if print_progress:
                    print("    LM update accepted, new residual: %g" % new)
                return True, float(np.float32(0.5) * lam), float(np.float32(x.max()))
            lam = np.float32(2.0) * lam
            if print_progress:
                print("    [%d of 10] LM update rejected (bad residual: %g), lambda increased to %g" % (lm + 1, new, lam))
            self.set_state(old)
        raise AssertionError("unreachable")

    def run_on_current_scale(self, max_num_iterations, max_change_convergence_threshold=0.0,
                             iterations_without_new_optimum_threshold=15, print_progress=False):
        self.current_image_scale = min(self.current_image_scale, self.image_scale_count - 2)
        if getattr(self, "cache_observations", False) and not hasattr(self, "observed"):
            self.determine_observed_indices()
        converged = False
        lam = np.float32(64.0)
        without = 0
        optimum_cost = float("inf")
        optimum = self.get_state()
        it = 0
        history = []
        while it < max_num_iterations:
            applied, max_change = True, float("inf")
            if it > 0:
                applied, lam, max_change = self.apply(lam, print_progress)
            self.update_observations(1)
            if self.var_weight > 0:
                self.color_update()
            cost = self.compute_cost()
            history.append(cost)
            if print_progress:
                print("  Cost (considering occlusions) is: %g" % cost)
            if cost < optimum_cost:
                optimum_cost = cost; without = 0; optimum = self.get_state()
            else:
                without += 1
            it += 1
            if (not applied) or max_change < max_change_convergence_threshold or without >= iterations_without_new_optimum_threshold:
                converged = True
                break
        self.set_state(optimum)
        self.history = history
        return converged, optimum_cost, it


# ---- `.observed_indices` files (observations_cache.cc:70-102 load, :127-158 save) -------------------------------------------------
def observed_indices_path(cache_dir, image_file_path):
    import os
    return os.path.join(cache_dir, os.path.basename(os.path.dirname(image_file_path)), os.path.basename(image_file_path) + ".observed_indices")


def write_observed_indices(path, lists):
    """lists: per point scale (ascending), arrays of point indices.  int32 count, then per scale u64 n + u64[n]."""
    import os
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(np.int32(len(lists)).tobytes())
        for l in lists:
            l = np.asarray(l, np.uint64)
            f.write(np.uint64(l.size).tobytes()); f.write(l.tobytes())


def read_observed_indices(path):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw, np.int32, 1, 0)[0]); pos = 4
    out = []
    for _ in range(n):
        c = int(np.frombuffer(raw, np.uint64, 1, pos)[0]); pos += 8
        out.append(np.frombuffer(raw, np.uint64, c, pos).copy()); pos += 8 * c
    return out
