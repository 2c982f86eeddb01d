"""TEST INFRASTRUCTURE: a second, separately written evaluator of the reference's first-hit + direct-light path, in numpy fp32.

It exists to pin the oracle (oracle/rt_oracle.cpp): the reference ships no fixtures and cannot be compiled here, so the
oracle's reading of the reference -- glm's operation order above all -- is checked against a second reading made
independently, straight from the reference's sources, with nothing shared but the scene arrays:

  ray generation        src/cpu/CPURenderer.cpp:111-116 (rand01), :158-192
  closest / any hit     src/cpu/CPUScene.cpp:107-137 (tree walk), :139-252 (Moeller-Trumbore on the pooled edges)
  box test              src/scene/AABB.h:47-60
  model tree            src/cpu/CPUScene.cpp:67-105
  pools                 src/scene/Scene.cpp:18-46, src/util/Model.cpp:3-26
  direct light          src/cpu/CPUScene.cpp:27-49, src/util/BRDF.cpp:4-14

Everything is vectorised over rays x triangles with numpy float32 arrays: each numpy operation rounds once to fp32, like
the scalar C++ (no FMA contraction), so the results can be compared bit for bit.  glm orders used (glm 0.9.9 / 1.0 scalar
paths): dot = (x*x' + y*y') + z*z'; cross = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y); normalize = v *
(1 / sqrt(dot(v, v))); mat4 * vec4 = (m0*x + m1*y) + (m2*z + m3*w); min(a, b) = b < a ? b : a; max(a, b) = a < b ? b : a.
render_first_hit evaluates depth 0 only (primary hit, its shadow rays, its lights: what a depth cap of 1 accumulates);
render_paths follows whole paths (src/cpu/CPUScene.cpp:6-65: roulette, emissive / throughput order, lights per depth, the
cosine-weighted bounce with its double-precision islands, the depth cap) and sums passes in `samp` order
(src/cpu/CPURenderer.cpp:196-197).  Its sin / cos are numpy's float64 functions rounded once to fp32 -- NOT the oracle's
sincos_det -- so the two share no transcendental code; a pixel where that choice matters would be reported by the test."""
import numpy as np

F = np.float32
FMAX = np.finfo(np.float32).max


def dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - b[..., 1] * a[..., 2],
                     a[..., 2] * b[..., 0] - b[..., 2] * a[..., 0],
                     a[..., 0] * b[..., 1] - b[..., 0] * a[..., 1]], -1)


def normalize(v):
    return v * (F(1.0) / np.sqrt(dot(v, v)))[..., None]


def vmin(a, b):  # glm::min / std::min
    return np.where(b < a, b, a)


def vmax(a, b):  # glm::max / std::max
    return np.where(a < b, b, a)


def rand01(w, h, samp):
    x = np.stack([w, h, samp], -1).astype(np.uint32)
    for _ in range(3):
        x = ((x >> np.uint32(8)) ^ x[..., [1, 2, 0]]) * np.uint32(1103515245)
    return x.astype(np.float32) * (F(1.0) / F(4294967295.0))


def primary_rays(cam, W, H, samp):
    """cam: dict(position, target, S_O, f, A, sensor) -> origins (H, W, 3), directions (H, W, 3), rnd (H, W, 3)"""
    pos, tgt = np.asarray(cam["position"], F), np.asarray(cam["target"], F)
    d = normalize(tgt - pos)
    up = np.array([0, 1, 0], F) if np.float64(abs(d[1])) < 0.9 else np.array([0, 0, 1], F)
    cx = normalize(cross(d, up))
    cy = cross(cx, d)
    sdim = np.asarray(cam.get("sensor", (0.036, 0.024)), F)
    S_O, f, A = F(cam["S_O"]), F(cam["f"]), F(cam["A"])
    S_I = (S_O * f) / (S_O - f)
    h, w = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    rnd = rand01(w, h, np.full_like(w, samp))
    rnd2 = F(2.0) * rnd[..., :2]
    with np.errstate(invalid="ignore"):
        tent = np.where(rnd2 < 1, np.sqrt(rnd2) - F(1), F(1) - np.sqrt(F(2) - rnd2))
    strat = np.array([(samp // 2) % 2, samp % 2], F)
    pix = np.stack([w, h], -1).astype(np.float32)
    s = ((pix + F(0.5) * ((F(0.5) + strat) + tent)) / np.array([W, H], F) - F(0.5)) * sdim
    spos = (pos + cx * s[..., 0:1]) + cy * s[..., 1:2]
    lc = pos + d * F(0.035)
    rdir = normalize(lc - spos)
    lensN = -d
    lensX = cross(lensN, np.array([0, 1, 0], F))
    lensY = cross(lensN, lensX)
    lens_sample = (lc + (rnd[..., 0:1] * A) * lensX) + (rnd[..., 1:2] * A) * lensY
    focal_point = pos + (S_O + S_I) * d
    t = dot(focal_point - lc, lensN) / dot(rdir, lensN)
    focus = lc + t[..., None] * rdir
    return lens_sample, normalize(focus - lens_sample), rnd


class Scene:
    """Scene::addModel + Scene::generate + CPUScene::createRayTracingHierarchy, from (positions, indices, 4x4 transform) triples
    (transforms in math convention: column vectors, as the tests write them)."""

    def __init__(self, models, reference_edge_indexing=True):
        """models: (positions, indices, transform) or (positions, indices, transform, texcoords, material) where material is
        None or a dict(kind, base_color, emissive, albedo_texture, emissive_texture) as Mesh.set_material takes it."""
        self.positions, self.indices, self.edges, self.normals, self.refs, self.boxes = [], [], [], [], [], []
        self.texcoords, self.materials = [], []
        pos_off = idx_off = 0
        for entry in models:
            p, idx, m = entry[:3]
            p = np.asarray(p, F).reshape(-1, 3)
            self.texcoords.append(np.asarray(entry[3], F).reshape(-1, 2) if len(entry) > 3 and entry[3] is not None else np.zeros((len(p), 2), F))
            self.materials.append(entry[4] if len(entry) > 4 else None)
            idx = np.asarray(idx, np.int64).reshape(-1, 3)
            m = np.asarray(m, F)
            cols = [m[:, k] for k in range(4)]  # glm columns
            lo, hi = p.min(0), p.max(0)          # AABB::adjust over the loaded positions
            # Model::transform: pos = vec3(matrix * vec4(pos, 1))
            tp = ((cols[0][:3] * p[:, 0:1] + cols[1][:3] * p[:, 1:2]) + (cols[2][:3] * p[:, 2:3] + cols[3][:3] * F(1)))
            corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], F)
            tc = ((cols[0][:3] * corners[:, 0:1] + cols[1][:3] * corners[:, 1:2]) + (cols[2][:3] * corners[:, 2:3] + cols[3][:3] * F(1)))
            self.boxes.append((tc.min(0), tc.max(0)))
            e0 = tp[idx[:, 1]] - tp[idx[:, 0]]
            e1 = tp[idx[:, 2]] - tp[idx[:, 0]]
            self.positions.append(tp)
            self.indices.append(idx)
            self.edges.append(np.stack([e0, e1], 1).reshape(-1, 3))  # two pool entries per triangle
            self.normals.append(normalize(cross(e0, e1)))
            self.refs.append((pos_off, len(tp), idx_off, len(idx)))
            pos_off += len(tp)
            idx_off += len(idx)
        self.position_pool = np.concatenate(self.positions)
        self.edges_pool = np.concatenate(self.edges)
        self.normal_pool = np.concatenate(self.normals)
        self.texcoord_pool = np.concatenate(self.texcoords)
        self.reference_edge_indexing = reference_edge_indexing
        self.tree = self._pair_models()

    # ---- HitInfo::texCoords (CPUScene.cpp:201-203, 217-219) and the material behind a hit
    def hit_texcoords(self, tri, o, d):
        """tri: pool triangle indices of the winners (R,), o / d their rays: the texture coordinates the intersector attaches"""
        uv = np.zeros((len(tri), 2), F)
        for m, (pos_off, _, idx_off, n) in enumerate(self.refs):
            sel = (tri >= idx_off) & (tri < idx_off + n)
            if not sel.any():
                continue
            local = tri[sel] - idx_off
            p0, e0, e1, _, _ = self._operands(m)
            idx = self.indices[m][local]
            t0, t1, t2 = (self.texcoord_pool[pos_off + idx[:, k]] for k in range(3))
            P0, E0, E1, O, D = p0[local], e0[local], e1[local], o[sel], d[sel]
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                sv = O - P0
                s1 = cross(D, E1)
                s2 = cross(sv, E0)
                frac = F(1.0) / dot(s1, E0)
                y = (dot(s1, sv) * frac)[:, None]
                z = (dot(s2, D) * frac)[:, None]
                b3 = F(1.0) - y - z
                uv[sel] = (t0 * y + t1 * z) + t2 * b3
        return uv

    @staticmethod
    def _sample(texture, uv):
        """Texture::sample (Texture.cpp:3-8); coordinates outside [0,1) clamp to the edge texels (the documented convention
        where the reference indexes out of bounds)"""
        tex = np.asarray(texture, F)
        h, w = tex.shape[:2]

        def index(c, size):
            f = c * F(size)
            with np.errstate(invalid="ignore"):
                inside = np.where(f >= F(size), size - 1, np.trunc(np.where(f >= 0, f, 0)).astype(np.int64))
            return np.where(f >= 0, inside, 0)
        return tex[index(uv[:, 1], h), index(uv[:, 0], w)]

    def evaluate_materials(self, tri, uv):
        """(albedo, emissive) per hit: Material::evaluate for kind 0, factor [* texture sample] for kind 1"""
        albedo = np.tile(np.array([0, 1, 0], F), (len(tri), 1))
        emissive = np.zeros((len(tri), 3), F)
        for m, (_, _, idx_off, n) in enumerate(self.refs):
            sel = (tri >= idx_off) & (tri < idx_off + n)
            spec = self.materials[m]
            if not sel.any() or spec is None:
                continue
            if spec["kind"] == 0:
                albedo[sel] = np.concatenate([uv[sel], np.zeros((sel.sum(), 1), F)], axis=1)
                emissive[sel] = 0
                continue
            a = np.tile(np.asarray(spec["base_color"], F), (sel.sum(), 1))
            e = np.tile(np.asarray(spec["emissive"], F), (sel.sum(), 1))
            if spec.get("albedo_texture") is not None:
                a = a * self._sample(spec["albedo_texture"], uv[sel])
            if spec.get("emissive_texture") is not None:
                e = e * self._sample(spec["emissive_texture"], uv[sel])
            albedo[sel], emissive[sel] = a, e
        return albedo, emissive

    def _pair_models(self):
        """CPUScene.cpp:67-105.  Returns nested tuples: ('leaf', model) | ('node', box, left, right)."""
        def area(b):
            d = b[1] - b[0]
            return F(2.0) * ((d[0] * d[1] + d[1] * d[2]) + d[2] * d[0])

        def combine(a, b):
            return (np.minimum(a[0], b[0]), np.maximum(a[1], b[1]))

        nodes = [(self.boxes[m], ("leaf", m)) for m in range(len(self.boxes))]
        while len(nodes) > 1:
            best, lhs, rhs = FMAX, 0, 0
            for i in range(len(nodes)):
                for j in range(len(nodes)):
                    if i == j:
                        continue
                    a = area(combine(nodes[i][0], nodes[j][0]))
                    if best > a:
                        best, lhs, rhs = a, i, j
            assert rhs > lhs
            box = combine(nodes[lhs][0], nodes[rhs][0])
            parent = (box, ("node", nodes[lhs], nodes[rhs]))
            del nodes[rhs]
            del nodes[lhs]
            nodes.append(parent)
        return nodes[0]

    # ---- AABB::intersects, vectorised over rays
    @staticmethod
    def box_hit(box, o, d, tmin, tmax):
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = F(1.0) / d
            t0 = (box[0] - o) * inv
            t1 = (box[1] - o) * inv
        sm, bg = vmin(t0, t1), vmax(t0, t1)
        lo = vmax(np.full(o.shape[:-1], tmin, F), vmax(sm[..., 0], vmax(sm[..., 1], sm[..., 2])))
        hi = vmin(np.full(o.shape[:-1], tmax, F), vmin(bg[..., 0], vmin(bg[..., 1], bg[..., 2])))
        return lo < hi

    def _operands(self, m):
        pos_off, _, idx_off, n = self.refs[m]
        idx = self.indices[m]
        p0 = self.position_pool[pos_off + idx[:, 0]]
        base = idx_off if self.reference_edge_indexing else 2 * idx_off  # CPUScene.cpp:157-158 reads indicesOffset + edgeIndex
        k = base + 2 * np.arange(n)
        return p0, self.edges_pool[k], self.edges_pool[k + 1], self.normal_pool[idx_off:idx_off + n], idx_off

    def _model_hits(self, m, o, d, tmin, tmax, any_hit, chunk=2048):
        """intersectModel / testModel over the rays (R, 3).  closest: returns (t, triangle index in pool or -1); any: bool"""
        p0, e0, e1, _, idx_off = self._operands(m)
        R = o.shape[0]
        best_t = np.full(R, FMAX, F)
        best_i = np.full(R, -1, np.int64)
        found = np.zeros(R, bool)
        for c0 in range(0, len(p0), chunk):
            P0, E0, E1 = p0[None, c0:c0 + chunk], e0[None, c0:c0 + chunk], e1[None, c0:c0 + chunk]
            O, D = o[:, None, :], d[:, None, :]
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                s = O - P0
                s1 = cross(np.broadcast_to(D, s.shape), np.broadcast_to(E1, s.shape))
                s2 = cross(s, np.broadcast_to(E0, s.shape))
                frac = F(1.0) / dot(s1, E0)
                t = dot(s2, E1) * frac
                b1 = dot(s1, s) * frac
                b2 = dot(s2, D) * frac
                b3 = F(1.0) - b1 - b2
                rej = (b3 < 0) | (b3 > 1) | (b1 < 0) | (b1 > 1) | (b2 < 0) | (b2 > 1) | (t < tmin) | (t > tmax)
            acc = ~rej
            if any_hit:
                found |= acc.any(1)
                continue
            # sequential overwrite rule (CPUScene.cpp:231): a triangle is skipped only if best.t < t, so among accepted
            # triangles the smallest t wins and equal t goes to the LATER index
            tt = np.where(acc, t, np.inf)
            cmin = tt.min(1)
            has = acc.any(1)
            last = tt.shape[1] - 1 - np.argmin(tt[:, ::-1], axis=1)  # last index attaining the chunk minimum
            take = has & ~(best_t < cmin)
            best_t = np.where(take, cmin.astype(F), best_t)
            best_i = np.where(take, idx_off + c0 + last, best_i)
        return found if any_hit else (best_t, best_i)

    def closest(self, o, d, tmin=F(1e-4), tmax=F(1e20)):
        """generateIntersections over the tree: (t, pool triangle index or -1) per ray"""
        def walk(node, mask):
            box, body = node
            R = o.shape[0]
            t = np.full(R, FMAX, F)
            i = np.full(R, -1, np.int64)
            m = mask & self.box_hit(box, o, d, tmin, tmax)
            if not m.any():
                return t, i
            if body[0] == "leaf":
                tt, ii = self._model_hits(body[1], o[m], d[m], tmin, tmax, False)
                t[m], i[m] = tt, ii
                return t, i
            lt, li = walk(body[1], m)
            rt, ri = walk(body[2], m)
            left = lt < rt  # CPUScene.cpp:136: right wins ties (and misses)
            return np.where(left, lt, rt), np.where(left, li, ri)
        return walk(self.tree, np.ones(o.shape[0], bool))

    def occluded(self, o, d, tmin=F(1e-4), tmax=F(1e20)):
        def walk(node, mask):
            box, body = node
            m = mask & self.box_hit(box, o, d, tmin, tmax)
            out = np.zeros(o.shape[0], bool)
            if not m.any():
                return out
            if body[0] == "leaf":
                out[m] = self._model_hits(body[1], o[m], d[m], tmin, tmax, True)
                return out
            return walk(body[1], m) | walk(body[2], m)
        return walk(self.tree, np.ones(o.shape[0], bool))


def render_first_hit(scene, cam, W, H, num_samples, samp, dir_lights, point_lights, materials=False, point_shadows=False):
    """One sample pass with a depth cap of 1: returns (accumulator contribution radiance/N (H, W, 3), triId (H, W), shadow bits
    (H, W), t (H, W)).  materials: albedo / emissive from the models' materials instead of CPUScene.cpp:243-247;
    point_shadows: point lights shadowed as in src/metal/Compute.metal:238-254."""
    o, d, _ = primary_rays(cam, W, H, samp)
    o, d = np.broadcast_to(o, d.shape).reshape(-1, 3).astype(F), d.reshape(-1, 3)
    t, tri = scene.closest(o, d)
    hit = t < FMAX
    radiance = np.zeros((W * H, 3), F)
    shadow = np.full(W * H, 0xFF, np.uint8)
    if hit.any():
        oh, dh, th, ih = o[hit], d[hit], t[hit], tri[hit]
        n = scene.normal_pool[ih]
        pos = oh + dh * th[:, None]
        albedo = np.tile(np.array([0, 1, 0], F), (len(th), 1))
        emissive = np.zeros((len(th), 3), F)
        if materials:
            albedo, emissive = scene.evaluate_materials(ih, scene.hit_texcoords(ih, oh, dh))
        rad = np.zeros((len(th), 3), F)
        rad = rad + (np.ones(3, F) * emissive) * F(1.0)  # CPUScene.cpp:27: material * emissive * payload.emissive
        bits = np.zeros(len(th), np.uint8)
        for l, (ldir, lcol) in enumerate(dir_lights):
            ldir, lcol = np.asarray(ldir, F), np.asarray(lcol, F)
            occ = scene.occluded(pos, np.broadcast_to(-ldir, pos.shape).astype(F))
            bits |= (occ.astype(np.uint8) << l)
            diffuse = vmax(dot(n, -ldir), F(0.0))
            e = (albedo * diffuse[:, None]) * lcol + F(0.03)
            rad = np.where(occ[:, None], rad, rad + e)
        for l, (lpos, lcol, att) in enumerate(point_lights):
            lpos, lcol = np.asarray(lpos, F), np.asarray(lcol, F)
            ldir = lpos - pos
            dist = np.sqrt(dot(ldir, ldir))
            illum = vmax(F(1) - dist / F(att), F(0.0))
            diffuse = vmax(dot(n, ldir), F(0.0))
            e = (albedo * diffuse[:, None]) * lcol + F(0.03)
            if point_shadows:
                # Compute.metal:240-246: from just off the surface toward the light, t in [0, 1]
                occ = scene.occluded((pos + n * F(1e-3)).astype(F), ldir.astype(F), tmin=F(0.0), tmax=F(1.0))
                bits |= (occ.astype(np.uint8) << (len(dir_lights) + l))
                rad = np.where(occ[:, None], rad, rad + illum[:, None] * e)
            else:
                rad = rad + illum[:, None] * e
        radiance[hit] = rad
        shadow[hit] = bits
    tri_id = np.where(hit, tri, 0xFFFFFFFF).astype(np.uint32)
    return (radiance / F(num_samples)).reshape(H, W, 3), tri_id.reshape(H, W), shadow.reshape(H, W), t.reshape(H, W)


def bounce_directions(normal_light, rnd):
    """CPUScene.cpp:54-60.  r1 is a double product narrowed to float (:54); |w.x| > 0.1 compares against a double literal (:58);
    cos / sin are the float overloads the file needs to compile at all: here float64 libm rounded once."""
    r1 = (2.0 * np.pi * rnd[:, 0].astype(np.float64)).astype(F)
    r2 = rnd[:, 1]
    r2s = np.sqrt(r2)
    w = normal_light
    pick_y = np.abs(w[:, 0]).astype(np.float64) > 0.1
    axis = np.where(pick_y[:, None], np.array([0, 1, 0], F), np.array([1, 0, 0], F))
    u = normalize(cross(axis, w))
    v = cross(w, u)
    c = np.cos(r1.astype(np.float64)).astype(F)
    sn = np.sin(r1.astype(np.float64)).astype(F)
    return normalize(((u * c[:, None]) * r2s[:, None] + (v * sn[:, None]) * r2s[:, None]) + w * np.sqrt(F(1) - r2)[:, None])


def render_paths(scene, cam, W, H, num_samples, samp, dir_lights, point_lights, max_depth=12, materials=False, point_shadows=False):
    """One sample pass of whole paths (traceRay, CPUScene.cpp:6-65, unrolled from its tail recursion).  Returns the pass's
    accumulator contribution radiance / N (H, W, 3), and per depth the hit ids (D, H, W) and shadow bits (D, H, W; 0xFF where
    the path has no hit at that depth)."""
    o, d, rnd = primary_rays(cam, W, H, samp)
    o, d, rnd = np.broadcast_to(o, d.shape).reshape(-1, 3).astype(F), d.reshape(-1, 3), rnd.reshape(-1, 3)
    R = W * H
    radiance = np.zeros((R, 3), F)     # payload.accumulatedRadiance
    material = np.ones((R, 3), F)      # payload.accumulatedMaterial
    emissive_on = F(1.0)               # payload.emissive: 1 for the camera ray's hit, 0 afterwards (:61)
    alive = np.arange(R)
    tri_aov = np.full((max_depth, R), 0xFFFFFFFF, np.uint32)
    shadow_aov = np.full((max_depth, R), 0xFF, np.uint8)
    for depth in range(max_depth):     # :15: a path that reaches depth 12 returns before adding anything
        if alive.size == 0:
            break
        t, tri = scene.closest(o, d)
        hit = t < FMAX                 # :10
        alive, o, d, rnd_a, t, tri = alive[hit], o[hit], d[hit], rnd[alive[hit]], t[hit], tri[hit]
        if alive.size == 0:
            break
        n = scene.normal_pool[tri]
        albedo = np.tile(np.array([0, 1, 0], F), (len(t), 1))   # :245-246
        emissive = np.zeros((len(t), 3), F)
        if materials:
            albedo, emissive = scene.evaluate_materials(tri, scene.hit_texcoords(tri, o, d))
        # :13-25 roulette (only beyond depth 5); p = max(max(x, y), z) with std::max's a < b ? b : a
        p = vmax(vmax(albedo[:, 0], albedo[:, 1]), albedo[:, 2])
        if depth > 5:
            keep = ~(rnd_a[:, 2] >= p)
            with np.errstate(divide="ignore", invalid="ignore"):
                material[alive[keep]] = material[alive[keep]] / p[keep, None]
            alive, o, d, rnd_a, t, tri, n, albedo, emissive = (x[keep] for x in (alive, o, d, rnd_a, t, tri, n, albedo, emissive))
            if alive.size == 0:
                break
        tri_aov[depth, alive] = tri   # (AOV convention shared with the oracle and the device: hits that get shaded)
        pos = o + d * t[:, None]       # :238
        rad = radiance[alive]
        mat = material[alive]
        rad = rad + (mat * emissive) * emissive_on   # :27
        mat = mat * albedo                           # :28
        bits = np.zeros(len(t), np.uint8)
        for l, (ldir, lcol) in enumerate(dir_lights):  # :31-38
            ldir, lcol = np.asarray(ldir, F), np.asarray(lcol, F)
            occ = scene.occluded(pos, np.broadcast_to(-ldir, pos.shape).astype(F))
            bits |= (occ.astype(np.uint8) << l)
            diffuse = vmax(dot(n, -ldir), F(0.0))
            e = (albedo * diffuse[:, None]) * lcol + F(0.03)   # BRDF.cpp:8,12 (unflipped normal, no throughput factor)
            rad = np.where(occ[:, None], rad, rad + e)
        for l, (lpos, lcol, att) in enumerate(point_lights):   # :39-49
            lpos, lcol = np.asarray(lpos, F), np.asarray(lcol, F)
            ldir = lpos - pos
            dist = np.sqrt(dot(ldir, ldir))
            illum = vmax(F(1) - dist / F(att), F(0.0))
            diffuse = vmax(dot(n, ldir), F(0.0))
            e = (albedo * diffuse[:, None]) * lcol + F(0.03)
            if point_shadows:  # Compute.metal:238-254
                occ = scene.occluded((pos + n * F(1e-3)).astype(F), ldir.astype(F), tmin=F(0.0), tmax=F(1.0))
                bits |= (occ.astype(np.uint8) << (len(dir_lights) + l))
                rad = np.where(occ[:, None], rad, rad + illum[:, None] * e)
            else:
                rad = rad + illum[:, None] * e
        shadow_aov[depth, alive] = bits
        radiance[alive] = rad
        material[alive] = mat
        # :54-63 the bounce: origin = hit position (no offset), same rnd01 at every depth
        with np.errstate(invalid="ignore"):
            normal_light = np.where((dot(n, d) < 0)[:, None], n, -n)   # :240
        d = bounce_directions(normal_light, rnd_a).astype(F)
        o = pos.astype(F)
        emissive_on = F(0.0)
    return ((radiance / F(num_samples)).reshape(H, W, 3), tri_aov.reshape(max_depth, H, W), shadow_aov.reshape(max_depth, H, W))
