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
// Poseidon kernel (the lane-cooperative one: a level is a few hundred hashes).
//
// WIDE levels stay on the device, NARROW ones go to the host.  A level costs the device one hash's dependency chain whatever its
// width (0.138 ms for hash4 on MI355X: ~400 dependent products of ~0.34 us on a GPU lane), while a host core runs the same chain in
// ~20 us: a group of at most STORE_HOST_MAX nodes is hashed by host_poseidon.hpp (the transcript's host permutation, any arity)
// instead.  The digest array lives on both sides; the ranges one side is missing are copied when a group changes sides (positions are
// assigned in (level, arity) order, so they are contiguous).  The 409-level list DAG of the tests - 25 wide levels of symbol
// hashing, then a 384-node spine - went from 74 ms to the figure DESIGN.md section 3.9 quotes; a wide DAG never leaves the device.
#include <algorithm>
#include <memory>

#include "common.hpp"
#include "field.cuh"
#include "host_poseidon.hpp"

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

// groups of at most this many nodes are hashed on the host: device level time / host hash time = 0.138 ms / ~20 us
constexpr uint32_t STORE_HOST_MAX = 6;

// host twin of store_gather_kernel + one hash: node k of the (level, arity)-ordered list from the host copy of the digest array
static void store_hash_node_host(int field_id, const NodeDev& nd, const uint64_t* dig, uint64_t* out4) {
    uint64_t pre[8 * 4] = {0};
    auto digest = [&](int e, uint32_t pos) { memcpy(pre + 4 * e, dig + 4 * (size_t)pos, 32); };
    auto tag = [&](int e, uint32_t t) { pre[4 * e] = t; };
    switch (nd.kind) {
        case LURK_NODE_TUPLE2:
        case LURK_NODE_TUPLE3:
        case LURK_NODE_TUPLE4:
            for (uint32_t e = 0; e < nd.arity; e++) {
                if (e & 1) digest((int)e, nd.child_pos[e >> 1]);
                else tag((int)e, nd.child_tag[e >> 1]);
            }
            break;
        case LURK_NODE_COMPACT:
            digest(0, nd.child_pos[0]);
            tag(1, nd.child_tag[1]);
            digest(2, nd.child_pos[1]);
            digest(3, nd.child_pos[2]);
            break;
        default:  // LURK_NODE_COMM
            digest(0, nd.secret_pos);
            tag(1, nd.child_tag[0]);
            digest(2, nd.child_pos[0]);
            break;
    }
    if (field_id == 0) poseidon_hash_host<PallasFp>(poseidon_host<PallasFp>((int)nd.arity), pre, out4);
    else if (field_id == 1) poseidon_hash_host<PallasFq>(poseidon_host<PallasFq>((int)nd.arity), pre, out4);
    else poseidon_hash_host<Bn254Fr>(poseidon_host<Bn254Fr>((int)nd.arity), pre, out4);
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
        // the digest array on both sides: positions below the two marks are valid on the host / on the device
        std::vector<uint64_t> host((n_values + order.size()) * 4);
        if (n_values) memcpy(host.data(), values32, n_values * 32);
        size_t host_valid = n_values, dev_valid = n_values;
        auto to_host = [&](size_t upto) {  // device-computed positions [host_valid, upto) come down
            if (upto <= host_valid) return;
            LURK_HIP_CHECK(hipMemcpyAsync(host.data() + host_valid * 4, (char*)d_dig.p + host_valid * 32, (upto - host_valid) * 32, hipMemcpyDeviceToHost, s));
            LURK_HIP_CHECK(hipStreamSynchronize(s));
            host_valid = upto;
        };
        auto to_device = [&](size_t upto) {  // host-computed positions [dev_valid, upto) go up
            if (upto <= dev_valid) return;
            LURK_HIP_CHECK(hipMemcpyAsync((char*)d_dig.p + dev_valid * 32, host.data() + dev_valid * 4, (upto - dev_valid) * 32, hipMemcpyHostToDevice, s));
            dev_valid = upto;
        };
        for (const Group& g : groups) {
            const size_t first = n_values + g.first, end = first + g.count;
            if (g.count <= STORE_HOST_MAX) {
                to_host(first);
                for (uint32_t k = 0; k < g.count; k++) store_hash_node_host(field_id, dev_nodes[g.first + k], host.data(), host.data() + (first + k) * 4);
                host_valid = end;
                continue;
            }
            to_device(first);
            ProfScope ps("store_hydrate_level", s);
            hipLaunchKernelGGL(store_gather_kernel, dim3(div_up((size_t)g.count * g.arity, 256)), dim3(256), 0, s, d_nodes.as<NodeDev>(), g.first, g.count,
                               d_dig.as<uint4>(), d_pre.as<uint4>());
            LURK_HIP_CHECK(hipGetLastError());
            poseidon_batch_device(field_id, g.arity, d_pre.p, (char*)d_dig.p + first * 32, g.count, 0, s);
            dev_valid = end;
            // (positions the host computed and no device group has needed yet stay below dev_valid: the marks only ever meet at a
            // group boundary, where whichever side is behind catches up with one copy)
        }
        to_host(n_values + order.size());
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < n; i++) memcpy((char*)digests32 + i * 32, host.data() + (size_t)pos[i] * 4, 32);
        if (levels_out) *levels_out = max_level;
    });
}

// pure host computation: usable without a device (the CPU tests run the reference's golden vectors through it)
extern "C" int lurk_hip_poseidon_hash_host(int field_id, int arity, const void* preimages, size_t n, void* digests) {
    try {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(arity == 3 || arity == 4 || arity == 6 || arity == 8, "unsupported arity (3, 4, 6 or 8)");  // src/hash.rs:19-29
        LURK_REQUIRE(n == 0 || (preimages && digests), "null buffer");
        for (size_t i = 0; i < n; i++) {
            uint64_t pre[8 * 4], out[4];
            memcpy(pre, (const char*)preimages + i * (size_t)arity * 32, (size_t)arity * 32);  // caller memory carries no alignment promise
            if (field_id == 0) poseidon_hash_host<PallasFp>(poseidon_host<PallasFp>(arity), pre, out);
            else if (field_id == 1) poseidon_hash_host<PallasFq>(poseidon_host<PallasFq>(arity), pre, out);
            else poseidon_hash_host<Bn254Fr>(poseidon_host<Bn254Fr>(arity), pre, out);
            memcpy((char*)digests + i * 32, out, 32);
        }
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}
