/*
 * osmt_geodata.hpp — host-side reader of the reference's on-disk geodata format: the scene
 * feed that replaces the synthetic generator.  Mirrors geodata::reader::GeodataReader (src/geodata/reader.rs)
 * over the layout written by geodata::saver (src/geodata/saver.rs:21-41,54-165), all little-endian:
 *
 *   u32 n_nodes   ; n x { u64 id, f64 lat, f64 lon, ref tags }                      (32 B, reader.rs:291)
 *   u32 n_ways    ; n x { u64 id, ref node_ids, ref tags }                          (24 B)
 *   u32 n_polys   ; n x { ref node_ids }                                            ( 8 B)
 *   u32 n_multis  ; n x { u64 id, ref polygon_ids, ref tags }                       (24 B)
 *   u32 n_tiles   ; n x { u32 x, u32 y, ref nodes, ref ways, ref multipolygons }    (32 B) z18 tiles, sorted by (x, y)
 *   u32 n_ints    ; n x u32                                                         (the pool every `ref` points into)
 *   string bytes                                                                    (to the end of the file)
 *   ref = { u32 offset, u32 length } into the int pool; tags = ints (k_off, k_len, v_off, v_len)* sorted by key.
 *
 * What the GPU path consumes: node_table() — every node's (lat, lon) packed [n][2] f64, uploaded ONCE — and the
 * node INDICES of ways / polygons, which go straight into osmt_batch.node_refs (OSMT_COORD_NODE_REF).  Styling
 * (which way becomes which op) stays with the integrator's styler.
 */
#ifndef OSMT_GEODATA_HPP
#define OSMT_GEODATA_HPP

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace osmt {

constexpr uint8_t GEODATA_MAX_ZOOM = 18; /* tile.rs:5 */

struct TileRange { /* tile.rs:15-21 */
    uint32_t min_x, max_x, min_y, max_y;
};

/* tile.rs:66-76 tile_to_max_zoom_tile_range */
inline TileRange tile_to_max_zoom_tile_range(uint8_t zoom, uint32_t x, uint32_t y) {
    const uint32_t f = 1u << (GEODATA_MAX_ZOOM - zoom);
    const uint32_t min_x = x * f, min_y = y * f, delta = f - 1u;
    return TileRange{min_x, min_x + delta, min_y, min_y + delta};
}

class GeodataReader;

/* reader.rs:339-398 Tags */
class Tags {
  public:
    Tags(const uint32_t* kv, size_t n_ints, const char* strings) : kv_(kv), n_(n_ints / 4), strings_(strings) {}
    size_t size() const { return n_; }
    std::pair<std::string_view, std::string_view> get_kv(size_t i) const {
        const uint32_t* r = kv_ + 4 * i;
        return {std::string_view(strings_ + r[0], r[1]), std::string_view(strings_ + r[2], r[3])};
    }
    /* binary search over keys (the saver writes a BTreeMap, i.e. sorted by key); reader.rs:353-376 */
    bool get_by_key(std::string_view key, std::string_view* value) const {
        if (n_ == 0) return false;
        size_t lo = 0, hi = n_ - 1;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const auto kv = get_kv(mid);
            const int c = kv.first.compare(key);
            if (c < 0)
                lo = mid + 1;
            else if (c > 0)
                hi = mid;
            else {
                *value = kv.second;
                return true;
            }
        }
        const auto kv = get_kv(lo);
        if (kv.first != key) return false;
        *value = kv.second;
        return true;
    }

  private:
    const uint32_t* kv_;
    size_t n_;
    const char* strings_;
};

struct OsmEntityIds { /* reader.rs:26-31 */
    std::vector<uint32_t> nodes, ways, multipolygons;
};

class GeodataReader {
  public:
    /* reader.rs:44-58 load: the file is memory-mapped, nothing is parsed up front */
    explicit GeodataReader(const std::string& path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("Failed to open " + path + " for memory mapping");
        struct stat st;
        if (fstat(fd_, &st) != 0 || st.st_size < 24) {
            ::close(fd_);
            throw std::runtime_error(path + " is not a geodata file");
        }
        len_ = (size_t)st.st_size;
        base_ = (const uint8_t*)mmap(nullptr, len_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (base_ == MAP_FAILED) {
            ::close(fd_);
            throw std::runtime_error("Failed to map " + path + " to memory");
        }
        const uint8_t* p = base_;
        const uint8_t* end = base_ + len_;
        auto storage = [&](Storage& s, size_t obj) { /* reader.rs:229-249 */
            if (p + 4 > end) throw std::runtime_error("truncated geodata file");
            s.count = rd32(p);
            s.size = obj;
            s.objects = p + 4;
            p = s.objects + obj * s.count;
            if (p > end) throw std::runtime_error("truncated geodata file");
        };
        storage(nodes_, 32);
        storage(ways_, 24);
        storage(polygons_, 8);
        storage(multipolygons_, 24);
        storage(tiles_, 32);
        if (p + 4 > end) throw std::runtime_error("truncated geodata file");
        n_ints_ = rd32(p);
        ints_ = reinterpret_cast<const uint32_t*>(p + 4); /* every member so far is a multiple of 4 bytes (reader.rs:276-278) */
        p += 4 + 4 * (size_t)n_ints_;
        if (p > end) throw std::runtime_error("truncated geodata file");
        strings_ = reinterpret_cast<const char*>(p);
    }
    ~GeodataReader() {
        if (base_ && base_ != MAP_FAILED) munmap(const_cast<uint8_t*>(base_), len_);
        if (fd_ >= 0) ::close(fd_);
    }
    GeodataReader(const GeodataReader&) = delete;
    GeodataReader& operator=(const GeodataReader&) = delete;

    size_t node_count() const { return nodes_.count; }
    size_t way_count() const { return ways_.count; }
    size_t polygon_count() const { return polygons_.count; }
    size_t multipolygon_count() const { return multipolygons_.count; }
    size_t tile_count() const { return tiles_.count; }

    /* Node (reader.rs:438-454) */
    uint64_t node_global_id(size_t i) const { return rd64(nodes_.at(i)); }
    double node_lat(size_t i) const { return rdf(nodes_.at(i) + 8); }
    double node_lon(size_t i) const { return rdf(nodes_.at(i) + 16); }
    Tags node_tags(size_t i) const { return tags(nodes_.at(i) + 24); }
    /* Way (reader.rs:456-472): node indices, usable directly as osmt_batch.node_refs */
    uint64_t way_global_id(size_t i) const { return rd64(ways_.at(i)); }
    std::pair<const uint32_t*, size_t> way_node_ids(size_t i) const { return ints_by_ref(ways_.at(i) + 8); }
    Tags way_tags(size_t i) const { return tags(ways_.at(i) + 16); }
    /* OsmArea::is_closed for ways (reader.rs:474-484) */
    bool way_is_closed(size_t i) const {
        const auto ids = way_node_ids(i);
        if (ids.second <= 2) return false;
        const uint32_t a = ids.first[0], b = ids.first[ids.second - 1];
        return node_lat(a) == node_lat(b) && node_lon(a) == node_lon(b);
    }
    /* Polygon (reader.rs:486-499) and Multipolygon (:501-517) */
    std::pair<const uint32_t*, size_t> polygon_node_ids(size_t i) const { return ints_by_ref(polygons_.at(i)); }
    uint64_t multipolygon_global_id(size_t i) const { return rd64(multipolygons_.at(i)); }
    std::pair<const uint32_t*, size_t> multipolygon_polygon_ids(size_t i) const { return ints_by_ref(multipolygons_.at(i) + 8); }
    Tags multipolygon_tags(size_t i) const { return tags(multipolygons_.at(i) + 16); }

    /* every node's (lat, lon), packed for osmt_batch.nodes: one upload serves every tile of the file */
    std::vector<double> node_table() const {
        std::vector<double> t(2 * nodes_.count);
        for (size_t i = 0; i < nodes_.count; ++i) {
            t[2 * i] = node_lat(i);
            t[2 * i + 1] = node_lon(i);
        }
        return t;
    }

    /* reader.rs:100-131 get_entities_in_tile: local ids of everything in the z18 tiles the tile covers, in file order
     * (x ascending, then y); duplicates are possible and are removed by the caller below */
    void get_entities_in_tile(uint8_t zoom, uint32_t x, uint32_t y, OsmEntityIds& out) const {
        TileRange bounds = tile_to_max_zoom_tile_range(zoom, x, y);
        size_t start_from_index = 0;
        const size_t tile_count = tiles_.count;
        while (start_from_index < tile_count) {
            size_t current_index;
            if (!next_good_tile(bounds, start_from_index, &current_index)) break;
            uint32_t tile_x = tile_x_at(current_index), tile_y = tile_y_at(current_index);
            const uint32_t current_x = tile_x;
            while (tile_x == current_x && tile_y <= bounds.max_y) {
                append(out.nodes, tiles_.at(current_index) + 8);
                append(out.ways, tiles_.at(current_index) + 16);
                append(out.multipolygons, tiles_.at(current_index) + 24);
                ++current_index;
                if (current_index >= tile_count) break;
                tile_x = tile_x_at(current_index);
                tile_y = tile_y_at(current_index);
            }
            start_from_index = current_index;
            bounds.min_x = current_x + 1;
        }
    }

    /* reader.rs:60-98 get_entities_in_tile_with_neighbors (without the osm_ids filter): the 3x3 neighbourhood,
     * sorted, unique; multipolygons without polygons are dropped */
    OsmEntityIds get_entities_in_tile_with_neighbors(uint8_t zoom, uint32_t x, uint32_t y) const {
        OsmEntityIds ids;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                get_entities_in_tile(zoom, (uint32_t)((int32_t)x + dx), (uint32_t)((int32_t)y + dy), ids);
        auto uniq = [](std::vector<uint32_t>& v) {
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        };
        uniq(ids.nodes);
        uniq(ids.ways);
        uniq(ids.multipolygons);
        ids.multipolygons.erase(std::remove_if(ids.multipolygons.begin(), ids.multipolygons.end(),
                                               [&](uint32_t m) { return multipolygon_polygon_ids(m).second == 0; }),
                                ids.multipolygons.end());
        return ids;
    }

  private:
    struct Storage { /* reader.rs:223-256 ObjectStorage */
        size_t count = 0, size = 0;
        const uint8_t* objects = nullptr;
        const uint8_t* at(size_t i) const { return objects + i * size; }
    };
    static uint32_t rd32(const uint8_t* p) {
        uint32_t v;
        memcpy(&v, p, 4);
        return v;
    }
    static uint64_t rd64(const uint8_t* p) {
        uint64_t v;
        memcpy(&v, p, 8);
        return v;
    }
    static double rdf(const uint8_t* p) {
        double v;
        memcpy(&v, p, 8);
        return v;
    }
    std::pair<const uint32_t*, size_t> ints_by_ref(const uint8_t* ref) const { /* reader.rs:209-214 */
        const uint32_t off = rd32(ref), len = rd32(ref + 4);
        if ((size_t)off + len > n_ints_) throw std::runtime_error("geodata reference out of range");
        return {ints_ + off, len};
    }
    Tags tags(const uint8_t* ref) const {
        const auto r = ints_by_ref(ref);
        return Tags(r.first, r.second, strings_);
    }
    void append(std::vector<uint32_t>& dst, const uint8_t* ref) const {
        const auto r = ints_by_ref(ref);
        dst.insert(dst.end(), r.first, r.first + r.second);
    }
    uint32_t tile_x_at(size_t i) const { return rd32(tiles_.at(i)); }
    uint32_t tile_y_at(size_t i) const { return rd32(tiles_.at(i) + 4); }
    bool large_enough(size_t idx, uint32_t min_x, uint32_t min_y) const { /* (x, y) >= (min_x, min_y), lexicographic */
        const uint32_t tx = tile_x_at(idx), ty = tile_y_at(idx);
        return tx > min_x || (tx == min_x && ty >= min_y);
    }
    /* reader.rs:133-177 next_good_tile */
    bool next_good_tile(TileRange& bounds, size_t start_index, size_t* out) const {
        const size_t tile_count = tiles_.count;
        if (start_index >= tile_count) return false;
        auto find_smallest_feasible_index = [&](size_t from, uint32_t min_x, uint32_t min_y, size_t* res) {
            size_t lo = from, hi = tile_count - 1;
            while (lo < hi) {
                const size_t mid = (lo + hi) / 2;
                if (large_enough(mid, min_x, min_y))
                    hi = mid;
                else
                    lo = mid + 1;
            }
            if (!large_enough(lo, min_x, min_y)) return false;
            *res = lo;
            return true;
        };
        size_t idx = start_index, next_idx;
        while (find_smallest_feasible_index(idx, bounds.min_x, bounds.min_y, &next_idx)) {
            const uint32_t tx = tile_x_at(next_idx), ty = tile_y_at(next_idx);
            if (tx > bounds.max_x || (tx == bounds.max_x && ty > bounds.max_y)) return false;
            if (tx == bounds.min_x) {
                *out = next_idx;
                return true;
            }
            idx = next_idx;
            bounds.min_x = tx;
        }
        return false;
    }

    int fd_ = -1;
    size_t len_ = 0;
    const uint8_t* base_ = nullptr;
    Storage nodes_, ways_, polygons_, multipolygons_, tiles_;
    const uint32_t* ints_ = nullptr;
    size_t n_ints_ = 0;
    const char* strings_ = nullptr;
};

}  // namespace osmt
#endif
