/* tools/model/tri_events.c -- study aid (not product, not a test): per-ray event traces of config 5's triangle path, for the wave
 * scheduling model tools/model/tri_wave_model.py.  Includes the oracle's source to reach its static functions.
 * Event stream of one ray, int16 each: 0 = a trip (an internal node entered: what one loop body of k_trace_lean_triangles does for
 * a lane), +k = a popped triangle leaf with k triangles, tested, no hit; -k = the same with a hit (the ray ends there).
 * Per pixel: primary stream, then the shadow stream (only when the colour kernel would trace it: primary hit and n.l > 0).
 *     gcc -O2 -ffp-contract=off -shared -fPIC -fopenmp -o build/libtri_events.so tools/model/tri_events.c -lm */
#include "../../oracle/rto_oracle.c"

static int trace_events(const orc_node* nodes, const float* tris, const int32_t* triOffset, const frame_consts* fc, v3 ro, v3 rd,
                        int16_t* ev, int cap, tri_result* res) {
    tri_result r; r.hit = 0; r.steps = 0; r.t = 1e30f; r.normal = v3_(0, 0, 0);
    int n = 0;
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    int traversalSteps = 0;
    v3 gmn = v3_(fc->gridMin[0], fc->gridMin[1], fc->gridMin[2]);
    float vs = fc->voxelSize;
    while (sp > 0 && traversalSteps < MAX_TRAVERSAL_STEPS) {
        sp--;
        int nodeIdx = stack[sp];
        if (nodeIdx < 0) continue;
        traversalSteps++;
        const orc_node* node = &nodes[nodeIdx];
        v3 nodeMin = v3_(gmn.x + (float)node->x * vs, gmn.y + (float)node->y * vs, gmn.z + (float)node->z * vs);
        float ext = (float)node->size * vs;
        v3 nodeMax = v3_(nodeMin.x + ext, nodeMin.y + ext, nodeMin.z + ext);
        float tNear, tFar;
        if (!intersect_aabb(ro, rd, nodeMin, nodeMax, &tNear, &tFar)) continue;
        if (tNear >= 1e30f) continue;
        if (node->isUniform == 1 || node->isLeaf == 1) {
            int cnt = triOffset[nodeIdx + 1] - triOffset[nodeIdx];
            if (cnt <= 0) continue;
            float bestT = 1e30f; int best = -1;
            for (int k = triOffset[nodeIdx]; k < triOffset[nodeIdx + 1]; k++) {
                float t;
                if (ray_triangle(ro, rd, tris + (size_t)k * 12, &t) && t < bestT) { bestT = t; best = k; }
            }
            int code = best >= 0 ? -cnt : cnt;
            if (best < 0) {
                /* would a box around the leaf's triangles, widened by 1e-3 voxel, have rejected the ray?  (study only: +1000 marks it) */
                v3 lo = v3_(1e30f, 1e30f, 1e30f), hi = v3_(-1e30f, -1e30f, -1e30f);
                for (int k = triOffset[nodeIdx]; k < triOffset[nodeIdx + 1]; k++)
                    for (int vtx = 0; vtx < 3; vtx++) {
                        const float* P3 = tris + (size_t)k * 12 + vtx * 3;
                        lo = v3_(gmin(lo.x, P3[0]), gmin(lo.y, P3[1]), gmin(lo.z, P3[2]));
                        hi = v3_(gmax(hi.x, P3[0]), gmax(hi.y, P3[1]), gmax(hi.z, P3[2]));
                    }
                float m = vs * 1e-3f, tn, tf;
                lo = v3_(lo.x - m, lo.y - m, lo.z - m); hi = v3_(hi.x + m, hi.y + m, hi.z + m);
                if (!intersect_aabb(ro, rd, lo, hi, &tn, &tf)) code += 1000;
            }
            if (n < cap) ev[n++] = (int16_t)code;
            if (best >= 0) {
                r.hit = 1; r.t = bestT;
                r.normal = v3_(tris[(size_t)best * 12 + 9], tris[(size_t)best * 12 + 10], tris[(size_t)best * 12 + 11]);
                break;
            }
            continue;
        }
        if (n < cap) ev[n++] = 0;
        for (int i = 0; i < 8; i++) {
            int childIdx = node->child[i];
            if (childIdx >= 0) stack[sp++] = childIdx;
        }
    }
    *res = r;
    return n;
}

/* tiles: nt (tx, ty) pairs of 8x8 tiles; out: per pixel of each tile 2 x cap int16 (primary, shadow), counts: 2 ints per pixel */
void tri_events_tiles(const orc_node* nodes, const float* tris, const int32_t* triOffset, const float gridMin[3], float voxelSize,
                      const float view[16], const float camPos[3], float aspect, float fovDeg, int W, int H,
                      const int32_t* tiles, int nt, int cap, int16_t* out, int32_t* counts) {
    frame_consts fc;
    frame_setup(&fc, gridMin, voxelSize, view, camPos, aspect, fovDeg, W, H);
    v3 ro = v3_(camPos[0], camPos[1], camPos[2]);
    v3 l = v3_normalize(v3_(-1.0f, -1.0f, -1.0f));
    v3 nl = v3_(-l.x, -l.y, -l.z);
    float bias = voxelSize * 1e-3f;
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < nt; t++)
        for (int lane = 0; lane < 64; lane++) {
            int px = tiles[2 * t] * 8 + (lane & 7), py = tiles[2 * t + 1] * 8 + (lane >> 3);
            size_t pix = (size_t)t * 64 + lane;
            counts[2 * pix] = counts[2 * pix + 1] = 0;
            if (px >= W || py >= H) continue;
            v3 rd;
            generate_ray(&fc, px, py, &rd);
            tri_result tr;
            counts[2 * pix] = trace_events(nodes, tris, triOffset, &fc, ro, rd, out + pix * 2 * cap, cap, &tr);
            if (!tr.hit) continue;
            v3 nrm = tr.normal;
            if (v3_dot(nrm, rd) > 0.0f) nrm = v3_(-nrm.x, -nrm.y, -nrm.z);
            float ndotl = gmax(0.0f, v3_dot(nrm, nl));
            if (!(ndotl > 0.0f)) continue;
            v3 p = v3_(ro.x + rd.x * tr.t, ro.y + rd.y * tr.t, ro.z + rd.z * tr.t);
            v3 so = v3_(p.x + nrm.x * bias, p.y + nrm.y * bias, p.z + nrm.z * bias);
            tri_result sh;
            counts[2 * pix + 1] = trace_events(nodes, tris, triOffset, &fc, so, nl, out + pix * 2 * cap + cap, cap, &sh);
        }
}
