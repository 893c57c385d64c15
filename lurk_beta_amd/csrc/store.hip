// store.hip - hydration of a content-addressed store DAG, every level on the device.
//
// Reference: StoreCore::hydrate_z_cache / hash_ptr (/root/reference/src/lem/store_core.rs:256-269) walks the DAG recursively
// and hashes node by node with the StoreHasher preimage layouts (/root/reference/src/lem/store.rs:29-78):
//   atom                 digest = the value itself                                  (store_core.rs:204)
//   tuple2 [a, b]        hash4(tag_a, h_a, tag_b, h_b)                              (store.rs:32-36)
//   tuple3 [a, b, c]     hash6(...)                                                 (store.rs:37-49)
//   tuple4 [a, b, c, d]  hash8(...)                                                 (store.rs:50-65)
//   compact [a, b, c]    hash4(h_a, tag_b, h_b, h_c)                                (store.rs:75-77)
//   comm (secret, a)     hash3(secret, tag_a, h_a)                                  (store.rs:70-73)
// Here the caller hands over the nodes in topological order (children before parents); the host only looks at the SHAPE
// (a node's level = 1 + max level of its children), groups the nodes by (level, arity), and the device does the rest: per
// group one gather kernel that builds the preimages from the children's tags and digests already in HBM and one batch of the
// Poseidon kernel (the lane-cooperative one: a level is a few hundred hashes).  Digests come back in one copy at the end;
// nothing crosses PCIe between levels.  Latency-bound by the DAG's depth (2 launches per level and arity).
#include <algorithm>
#include <memory>

#include "common.hpp"
#include "field.cuh"

namespace lurk {

void poseidon_batch_device(int field_id, int arity, const void* d_pre, void* d_out, size_t n, int flags, hipStream_t s);  // poseidon.hip

struct NodeDev {  // device copy of a node, children already translated to digest-array positions
    uint32_t kind, arity;
    uint32_t child_pos[4];
    uint32_t child_tag[4];
    uint32_t secret_pos;  // comm: position of the secret value in the digest array
};

// one thread per (group node, preimage element): canonical 32-byte elements
__global__ __launch_bounds__(256) void store_gather_kernel(const NodeDev* __restrict__ nodes, uint32_t first, uint32_t count, const uint4* __restrict__ digests,
                                                             uint4* __restrict__ pre) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const NodeDev nd0 = nodes[first];
    const uint32_t arity = nd0.arity;
    if (t >= count * arity) return;
    const uint32_t i = t / arity, e = t % arity;
    const NodeDev nd = nodes[first + i];
    uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
    auto digest = [&](uint32_t pos) { lo = digests[2 * (size_t)pos]; hi = digests[2 * (size_t)pos + 1]; };
    switch (nd.kind) {
        case LURK_NODE_TUPLE2:
        case LURK_NODE_TUPLE3:
        case LURK_NODE_TUPLE4:  // [tag, hash] per child
            if (e & 1) digest(nd.child_pos[e >> 1]);
            else lo.x = nd.child_tag[e >> 1];
            break;
        case LURK_NODE_COMPACT:  // h_a, tag_b, h_b, h_c
            if (e == 0) digest(nd.child_pos[0]);
            else if (e == 1) lo.x = nd.child_tag[1];
            else digest(nd.child_pos[e - 1]);
            break;
        default:  // LURK_NODE_COMM: secret, tag_a, h_a
            if (e == 0) digest(nd.secret_pos);
            else if (e == 1) lo.x = nd.child_tag[0];
            else digest(nd.child_pos[0]);
            break;
    }
    pre[2 * (size_t)t] = lo;
    pre[2 * (size_t)t + 1] = hi;
}

static int kind_arity(uint32_t kind) {
    switch (kind) {
        case LURK_NODE_TUPLE2: return 4;
        case LURK_NODE_TUPLE3: return 6;
        case LURK_NODE_TUPLE4: return 8;
        case LURK_NODE_COMPACT: return 4;
        case LURK_NODE_COMM: return 3;
        default: return 0;
    }
}
static int kind_children(uint32_t kind) {
    switch (kind) {
        case LURK_NODE_TUPLE2: return 2;
        case LURK_NODE_TUPLE3: return 3;
        case LURK_NODE_TUPLE4: return 4;
        case LURK_NODE_COMPACT: return 3;
        case LURK_NODE_COMM: return 1;
        default: return 0;
    }
}

}  // namespace lurk

using namespace lurk;

extern "C" int lurk_hip_store_hydrate(int field_id, const lurk_hip_store_node* nodes, size_t n, const void* values32, size_t n_values, void* digests32,
                                      size_t* levels_out) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(n == 0 || (nodes && digests32), "null buffer");
        LURK_REQUIRE(n < ((size_t)1 << 31) && n_values < ((size_t)1 << 31), "too many nodes");
        if (levels_out) *levels_out = 0;
        if (n == 0) return;
        // ---- shape: levels, then positions in the device digest array = [values | nodes grouped by (level, arity)] ----
        std::vector<uint32_t> level(n, 0);
        uint32_t max_level = 0;
        for (size_t i = 0; i < n; i++) {
            const lurk_hip_store_node& nd = nodes[i];
            if (nd.kind == LURK_NODE_ATOM) {
                LURK_REQUIRE(nd.value < n_values && values32, "atom value index out of range");
                continue;
            }
            const int nc = kind_children(nd.kind);
            LURK_REQUIRE(nc > 0, "unknown node kind");
            uint32_t lv = 0;
            for (int k = 0; k < nc; k++) {
                LURK_REQUIRE(nd.child[k] < i, "nodes must be in topological order (children first)");
                lv = std::max(lv, level[nd.child[k]]);
            }
            if (nd.kind == LURK_NODE_COMM) LURK_REQUIRE(nd.value < n_values && values32, "commitment secret index out of range");
            level[i] = lv + 1;
            max_level = std::max(max_level, lv + 1);
        }
        std::vector<uint32_t> order;  // hashed nodes sorted by (level, arity), stable
        order.reserve(n);
        for (size_t i = 0; i < n; i++)
            if (nodes[i].kind != LURK_NODE_ATOM) order.push_back((uint32_t)i);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            if (level[a] != level[b]) return level[a] < level[b];
            return kind_arity(nodes[a].kind) < kind_arity(nodes[b].kind);
        });
        std::vector<uint32_t> pos(n);
        for (size_t i = 0; i < n; i++)
            if (nodes[i].kind == LURK_NODE_ATOM) pos[i] = nodes[i].value;
        for (size_t k = 0; k < order.size(); k++) pos[order[k]] = (uint32_t)(n_values + k);
        std::vector<NodeDev> dev_nodes(order.size());
        size_t max_group_elems = 0;
        struct Group { uint32_t first, count; int arity; };
        std::vector<Group> groups;
        for (size_t k = 0; k < order.size(); k++) {
            const lurk_hip_store_node& nd = nodes[order[k]];
            NodeDev& d = dev_nodes[k];
            d.kind = nd.kind;
            d.arity = (uint32_t)kind_arity(nd.kind);
            for (int c = 0; c < 4; c++) {
                const bool used = c < kind_children(nd.kind);
                d.child_pos[c] = used ? pos[nd.child[c]] : 0;
                d.child_tag[c] = used ? nodes[nd.child[c]].tag : 0;
            }
            d.secret_pos = nd.kind == LURK_NODE_COMM ? nd.value : 0;
            if (groups.empty() || level[order[k]] != level[order[groups.back().first]] || (int)d.arity != groups.back().arity)
                groups.push_back({(uint32_t)k, 0, (int)d.arity});
            groups.back().count++;
            max_group_elems = std::max(max_group_elems, (size_t)groups.back().count * d.arity);
        }
        // ---- device: values + digests resident, one gather + one hash batch per group, one copy back ----
        hipStream_t s = nullptr;
        DevBuf d_nodes(dev_nodes.size() * sizeof(NodeDev)), d_dig((n_values + order.size()) * 32), d_pre(max_group_elems * 32);
        if (!dev_nodes.empty()) LURK_HIP_CHECK(hipMemcpyAsync(d_nodes.p, dev_nodes.data(), dev_nodes.size() * sizeof(NodeDev), hipMemcpyHostToDevice, s));
        if (n_values) LURK_HIP_CHECK(hipMemcpyAsync(d_dig.p, values32, n_values * 32, hipMemcpyHostToDevice, s));
        for (const Group& g : groups) {
            ProfScope ps("store_hydrate_level", s);
            hipLaunchKernelGGL(store_gather_kernel, dim3(div_up((size_t)g.count * g.arity, 256)), dim3(256), 0, s, d_nodes.as<NodeDev>(), g.first, g.count,
                               d_dig.as<uint4>(), d_pre.as<uint4>());
            LURK_HIP_CHECK(hipGetLastError());
            poseidon_batch_device(field_id, g.arity, d_pre.p, (char*)d_dig.p + (n_values + g.first) * 32, g.count, 0, s);
        }
        std::vector<uint64_t> host((n_values + order.size()) * 4);
        LURK_HIP_CHECK(hipMemcpyAsync(host.data(), d_dig.p, host.size() * 8, hipMemcpyDeviceToHost, s));
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < n; i++) memcpy((char*)digests32 + i * 32, host.data() + (size_t)pos[i] * 4, 32);
        if (levels_out) *levels_out = max_level;
    });
}
