// binary_io.hip -- reading and writing the reference's .rebx parameter files (host code only).
//
// Format (reference src/output.c:31-86, src/reboundx.h:99-127, :224-227): a 64-byte text header,
// then nested records { int32 type; 4 bytes padding; int64 size; payload[size] }.  Container
// records (snapshot, rebx structure, the lists, force, operator, step, particle, param) hold
// child records and close with an END record that is counted in their size; leaf records hold raw
// bytes.  Lists are stored oldest element first, so that re-inserting at the head while reading
// restores the original list order (src/output.c:209-256).
//
// The whole file is read into memory and parsed with bounds-checked cursors; writing builds the
// file in a byte buffer and patches each container's size when it closes.
#include "../../include/reboundx_amd.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" const char* rebx_version_str;
extern "C" const char* rebx_githash_str;

namespace {

enum FieldType : int32_t {            // src/reboundx.h:99-127
    F_NONE = 0, F_OPERATOR = 1, F_PARTICLE = 2, F_REBX_STRUCTURE = 3, F_PARAM = 4, F_NAME = 5, F_PARAM_TYPE = 6, F_PARAM_VALUE = 7,
    F_END = 8, F_PARTICLE_INDEX = 9, F_STEP = 13, F_STEP_DT_FRACTION = 14, F_REGISTERED_PARAM = 15, F_ADDITIONAL_FORCE = 16,
    F_PARAM_LIST = 17, F_REGISTERED_PARAMETERS = 18, F_ALLOCATED_FORCES = 19, F_ALLOCATED_OPERATORS = 20, F_ADDITIONAL_FORCES = 21,
    F_PRE_TIMESTEP_MODIFICATIONS = 22, F_POST_TIMESTEP_MODIFICATIONS = 23, F_PARTICLES = 24, F_FORCE = 25, F_SNAPSHOT = 26
};
constexpr size_t HEADER = 64, RECORD = 16;

// ---------------------------------------------------------------- writer
struct Writer {
    std::string buf;
    void record(int32_t type, const void* data, size_t size) {
        char head[RECORD] = {0};
        const int64_t s = (int64_t)size;
        memcpy(head, &type, 4);
        memcpy(head + 8, &s, 8);
        buf.append(head, RECORD);
        if (size) buf.append((const char*)data, size);
    }
    void text(int32_t type, const char* s) { record(type, s, strlen(s) + 1); }
    size_t open(int32_t type) { record(type, nullptr, 0); return buf.size(); }
    void close(size_t payload_start) {
        record(F_END, nullptr, 0);
        const int64_t s = (int64_t)(buf.size() - payload_start);
        memcpy(&buf[payload_start - 8], &s, 8);
    }
};

size_t value_size(enum rebx_param_type t) {
    switch (t) {
    case REBX_TYPE_DOUBLE: return sizeof(double);
    case REBX_TYPE_INT: return sizeof(int);
    case REBX_TYPE_UINT32: return sizeof(uint32_t);      // the reference's rebx_sizeof has no case for it (core.c:968-1001) and writes 0 bytes
    default: return 0;
    }
}

// oldest first: the list head is the most recently added element
template <class Fn> void for_each_oldest_first(struct rebx_node* head, Fn fn) {
    std::vector<struct rebx_node*> nodes;
    for (struct rebx_node* n = head; n; n = n->next) nodes.push_back(n);
    for (size_t k = nodes.size(); k-- > 0;) fn(nodes[k]->object);
}

// Per-particle double parameters that were given as whole columns (rebxamd_set_param_column) have no list node; a file
// holds them as what they stand for, a parameter of the particle, so that the reference (and this library) reads them back
// as list entries.
struct ColumnEntries {
    static constexpr int NCOL = 9;
    const char* name[NCOL] = {"beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc", "tau_mass", "tau_omega", "tau_Omega"};
    const double* col[NCOL] = {};
    int len[NCOL] = {};
    bool any = false;
    explicit ColumnEntries(struct rebx_extras* rebx) {
        for (int k = 0; k < NCOL; k++) {
            col[k] = rebxamd_get_param_column(rebx, name[k], &len[k]);
            any = any || col[k] != nullptr;
        }
    }
};

void write_param_list(Writer& w, struct rebx_extras* rebx, struct rebx_node* ap, const ColumnEntries* columns = nullptr, int particle = -1) {
    const size_t list = w.open(F_PARAM_LIST);
    if (columns != nullptr && columns->any) {
        // (written first = oldest: entries of the particle's own list end up in front of them when the file is read)
        for (int k = 0; k < ColumnEntries::NCOL; k++) {
            if (!columns->col[k] || particle >= columns->len[k]) continue;
            const double value = columns->col[k][particle];
            if (value != value || rebx_get_param(rebx, ap, columns->name[k]) != NULL) continue;     // not set here / the list's own entry wins
            const enum rebx_param_type type = REBX_TYPE_DOUBLE;
            const size_t rec = w.open(F_PARAM);
            w.record(F_PARAM_TYPE, &type, sizeof(type));
            w.text(F_NAME, columns->name[k]);
            w.record(F_PARAM_VALUE, &value, sizeof(value));
            w.close(rec);
        }
    }
    for_each_oldest_first(ap, [&](void* obj) {
        const struct rebx_param* p = (const struct rebx_param*)obj;
        if (p->type == REBX_TYPE_POINTER) return;        // cannot be restored (output.c:120-122)
        const size_t rec = w.open(F_PARAM);
        w.record(F_PARAM_TYPE, &p->type, sizeof(p->type));
        w.text(F_NAME, p->name);
        if (p->type == REBX_TYPE_FORCE) w.text(F_PARAM_VALUE, ((const struct rebx_force*)p->value)->name);   // by name (output.c:110-118)
        else w.record(F_PARAM_VALUE, p->value, value_size(p->type));
        w.close(rec);
    });
    w.close(list);
}

void write_named_with_params(Writer& w, struct rebx_extras* rebx, int32_t type, const char* name, struct rebx_node* ap) {
    const size_t rec = w.open(type);
    w.text(F_NAME, name);
    write_param_list(w, rebx, ap);
    w.close(rec);
}

void write_steps(Writer& w, int32_t list_type, struct rebx_node* head) {
    const size_t list = w.open(list_type);
    for_each_oldest_first(head, [&](void* obj) {
        const struct rebx_step* s = (const struct rebx_step*)obj;
        const size_t rec = w.open(F_STEP);
        w.text(F_NAME, s->op->name);
        w.record(F_STEP_DT_FRACTION, &s->dt_fraction, sizeof(s->dt_fraction));
        w.close(rec);
    });
    w.close(list);
}

// ---------------------------------------------------------------- reader
struct Cursor {
    const unsigned char* p;
    const unsigned char* end;
};
struct Record { int32_t type; int64_t size; const unsigned char* payload; };

bool next_record(Cursor& c, Record& r) {
    if ((size_t)(c.end - c.p) < RECORD) return false;
    memcpy(&r.type, c.p, 4);
    memcpy(&r.size, c.p + 8, 8);
    r.payload = c.p + RECORD;
    c.p += RECORD;
    return true;
}
// payload of a leaf record, bounds-checked; advances past it
bool take(Cursor& c, const Record& r, const unsigned char*& data) {
    if (r.size < 0 || r.size > c.end - c.p) return false;
    data = c.p;
    c.p += r.size;
    return true;
}
void skip(Cursor& c, const Record& r) {
    if (r.size < 0 || r.size > c.end - c.p) c.p = c.end; else c.p += r.size;
}

struct Loader {
    struct rebx_extras* rebx;
    int warnings = 0;

    char* dup_bytes(const unsigned char* d, int64_t n) {
        char* out = (char*)malloc(n > 0 ? (size_t)n : 1);
        if (!out) { warnings |= REBX_INPUT_BINARY_ERROR_NO_MEMORY; return nullptr; }
        if (n > 0) memcpy(out, d, (size_t)n);
        return out;
    }
    bool push(struct rebx_node** head, void* object) {
        struct rebx_node* n = (struct rebx_node*)malloc(sizeof(*n));
        if (!n) { warnings |= REBX_INPUT_BINARY_ERROR_NO_MEMORY; return false; }
        n->object = object; n->next = *head; *head = n;
        return true;
    }
    static void drop(struct rebx_param* p) { if (p) { free(p->name); free(p->value); free(p); } }

    // PARAM and REGISTERED_PARAM payloads (input.c:62-104); nullptr when unusable
    int64_t value_bytes = 0;            // size of the PARAM_VALUE payload of the parameter read last

    struct rebx_param* read_param(Cursor& c) {
        value_bytes = 0;
        struct rebx_param* p = (struct rebx_param*)calloc(1, sizeof(*p));
        if (!p) { warnings |= REBX_INPUT_BINARY_ERROR_NO_MEMORY; return nullptr; }
        p->type = REBX_TYPE_NONE;
        Record r;
        bool closed = false;
        while (next_record(c, r)) {
            if (r.type == F_END) { closed = true; break; }
            const unsigned char* d = nullptr;
            if (r.type == F_PARAM_TYPE || r.type == F_NAME || r.type == F_PARAM_VALUE) {
                if (!take(c, r, d)) { warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; break; }
                if (r.type == F_PARAM_TYPE) { int32_t t = 0; memcpy(&t, d, r.size < 4 ? (size_t)r.size : 4); p->type = (enum rebx_param_type)t; }
                else if (r.type == F_NAME) { free(p->name); p->name = dup_bytes(d, r.size); if (p->name && r.size > 0) p->name[r.size - 1] = '\0'; }
                else if (r.size > 0) { free(p->value); p->value = dup_bytes(d, r.size); value_bytes = p->value ? r.size : 0; }
                else warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;       // an empty value reads as a failed fread there
            } else {
                warnings |= REBX_INPUT_BINARY_WARNING_FIELD_UNKNOWN;
            }
        }
        if (!closed) warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;
        if (p->type == REBX_TYPE_NONE || p->name == nullptr) { warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; drop(p); return nullptr; }
        return p;
    }

    bool load_param(Cursor& c, struct rebx_node** ap) {                 // input.c:106-135
        struct rebx_param* p = read_param(c);
        if (!p) return false;
        if (!p->value) { warnings |= REBX_INPUT_BINARY_WARNING_PARAM_VALUE_NULL; drop(p); return false; }
        // The payload's size is the file's claim, the type says what readers of the value will touch:
        // a number must have exactly its type's size, a force name must end inside the payload, and a
        // value of a type nobody here can own or free (pointer, orbit, unknown) is not kept.
        if (p->type == REBX_TYPE_FORCE) {
            ((char*)p->value)[value_bytes - 1] = '\0';
        } else if (value_size(p->type) == 0) {
            warnings |= REBX_INPUT_BINARY_WARNING_PARAM_VALUE_NULL; drop(p); return false;
        } else if ((size_t)value_bytes != value_size(p->type)) {
            warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; drop(p); return false;
        }
        if (p->type == REBX_TYPE_FORCE) {
            struct rebx_force* f = rebx_get_force(rebx, (const char*)p->value);
            if (!f) { warnings |= REBX_INPUT_BINARY_WARNING_FORCE_PARAM_NOT_LOADED; drop(p); return false; }
            free(p->value);
            p->value = f;
        }
        if (!push(ap, p)) { if (p->type == REBX_TYPE_FORCE) p->value = nullptr; drop(p); return false; }
        return true;
    }

    bool read_name(Cursor& c, std::string& name) {                      // input.c:152-175
        Record r;
        const unsigned char* d = nullptr;
        if (!next_record(c, r) || r.type != F_NAME || !take(c, r, d) || r.size <= 0) { warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; return false; }
        name.assign((const char*)d, strnlen((const char*)d, (size_t)r.size));
        return true;
    }

    // children of a force / operator / particle record after its leading field: PARAM_LIST ... END
    bool read_params_until_end(Cursor& c, struct rebx_node** ap) {
        Record r;
        while (next_record(c, r)) {
            if (r.type == F_END) return true;
            if (r.type == F_PARAM_LIST) {
                Cursor inner{c.p, c.p + (r.size >= 0 && r.size <= c.end - c.p ? r.size : 0)};
                if (!load_list(inner, F_PARAM, ap)) return false;
                skip(c, r);
            } else {
                warnings |= REBX_INPUT_BINARY_WARNING_FIELD_UNKNOWN;
                skip(c, r);
            }
        }
        warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;
        return false;
    }

    bool load_force(Cursor& c) {                                        // input.c:177-214
        std::string name;
        if (!read_name(c, name)) return false;
        struct rebx_force* f = rebx_load_force(rebx, name.c_str());
        if (!f) { warnings |= REBX_INPUT_BINARY_WARNING_FORCE_NOT_LOADED; return false; }
        return read_params_until_end(c, &f->ap);
    }
    bool load_operator(Cursor& c) {                                     // input.c:253-295
        std::string name;
        if (!read_name(c, name)) return false;
        struct rebx_operator* op = rebx_load_operator(rebx, name.c_str());
        if (!op) { warnings |= REBX_INPUT_BINARY_WARNING_OPERATOR_NOT_LOADED; return false; }
        read_params_until_end(c, &op->ap);
        return true;
    }
    bool load_additional_force(Cursor& c) {                             // input.c:217-251
        std::string name;
        if (!read_name(c, name)) return false;
        struct rebx_force* f = rebx_get_force(rebx, name.c_str());
        if (!f) return false;
        Record r;
        bool closed = false;
        while (next_record(c, r)) {
            if (r.type == F_END) { closed = true; break; }
            warnings |= REBX_INPUT_BINARY_WARNING_FIELD_UNKNOWN;
            skip(c, r);
        }
        if (!closed) { warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; return false; }
        return rebx_add_force(rebx, f) != 0;
    }
    bool load_step(Cursor& c, struct rebx_node** list) {                // input.c:297-345
        std::string name;
        if (!read_name(c, name)) return false;
        struct rebx_operator* op = rebx_get_operator(rebx, name.c_str());
        if (!op) { warnings |= REBX_INPUT_BINARY_WARNING_OPERATOR_NOT_LOADED; return false; }
        double fraction = 0.;
        Record r;
        bool closed = false;
        while (next_record(c, r)) {
            if (r.type == F_END) { closed = true; break; }
            const unsigned char* d = nullptr;
            if (r.type == F_STEP_DT_FRACTION && r.size == (int64_t)sizeof(double) && take(c, r, d)) memcpy(&fraction, d, sizeof(double));
            else { warnings |= REBX_INPUT_BINARY_WARNING_FIELD_UNKNOWN; skip(c, r); }
        }
        if (!closed) warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;
        if (fraction == 0.) return false;
        if (list == &rebx->pre_timestep_modifications) return rebx_add_operator_step(rebx, op, fraction, REBX_TIMING_PRE) != 0;
        if (list == &rebx->post_timestep_modifications) return rebx_add_operator_step(rebx, op, fraction, REBX_TIMING_POST) != 0;
        return false;
    }
    bool load_particle(Cursor& c) {                                     // input.c:347-396
        Record r;
        const unsigned char* d = nullptr;
        if (!next_record(c, r) || r.type != F_PARTICLE_INDEX || r.size != 4 || !take(c, r, d)) { warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; return false; }
        int32_t index = 0;
        memcpy(&index, d, 4);
        struct reb_simulation* sim = rebx->sim;
        if (index < 0 || index >= sim->N) { warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; return false; }   // the reference indexes unchecked
        return read_params_until_end(c, (struct rebx_node**)&sim->particles[index].ap);
    }

    // a list container: records of `expected` type until END (input.c:531-614).  false = wrong format.
    bool load_list(Cursor& c, int32_t expected, struct rebx_node** ap) {
        Record r;
        while (next_record(c, r)) {
            if (r.type == F_END) return true;
            if (r.type != expected) return false;
            Cursor inner{c.p, c.p + (r.size >= 0 && r.size <= c.end - c.p ? r.size : 0)};
            switch (expected) {
            case F_PARAM: if (!load_param(inner, ap)) warnings |= REBX_INPUT_BINARY_WARNING_PARAM_NOT_LOADED; break;
            case F_REGISTERED_PARAM: {
                struct rebx_param* p = read_param(inner);
                if (!p || !push(&rebx->registered_params, p)) { drop(p); warnings |= REBX_INPUT_BINARY_ERROR_REGISTERED_PARAM_NOT_LOADED; }
                break;
            }
            case F_FORCE: if (!load_force(inner)) warnings |= REBX_INPUT_BINARY_WARNING_FORCE_NOT_LOADED; break;
            case F_ADDITIONAL_FORCE: if (!load_additional_force(inner)) warnings |= REBX_INPUT_BINARY_WARNING_ADDITIONAL_FORCE_NOT_LOADED; break;
            case F_OPERATOR: if (!load_operator(inner)) warnings |= REBX_INPUT_BINARY_WARNING_OPERATOR_NOT_LOADED; break;
            case F_STEP: if (!load_step(inner, ap)) warnings |= REBX_INPUT_BINARY_WARNING_STEP_NOT_LOADED; break;
            case F_PARTICLE: if (!load_particle(inner)) warnings |= REBX_INPUT_BINARY_WARNING_PARTICLE_PARAMS_NOT_LOADED; break;
            default: return false;
            }
            skip(c, r);
        }
        return false;           // ran off the end without END
    }

    void load_structure(Cursor& c) {                                    // input.c:398-470
        static const struct { int32_t list, item; int which; } lists[] = {
            {F_REGISTERED_PARAMETERS, F_REGISTERED_PARAM, 0}, {F_ALLOCATED_FORCES, F_FORCE, 1}, {F_ALLOCATED_OPERATORS, F_OPERATOR, 1},
            {F_ADDITIONAL_FORCES, F_ADDITIONAL_FORCE, 1}, {F_PRE_TIMESTEP_MODIFICATIONS, F_STEP, 2}, {F_POST_TIMESTEP_MODIFICATIONS, F_STEP, 3}};
        Record r;
        while (next_record(c, r)) {
            if (r.type == F_END) return;
            bool known = false;
            for (const auto& l : lists) {
                if (r.type != l.list) continue;
                known = true;
                struct rebx_node** head = l.which == 0 ? &rebx->registered_params : l.which == 2 ? &rebx->pre_timestep_modifications
                                        : l.which == 3 ? &rebx->post_timestep_modifications : nullptr;
                Cursor inner{c.p, c.p + (r.size >= 0 && r.size <= c.end - c.p ? r.size : 0)};
                if (!load_list(inner, l.item, head)) warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;
            }
            if (!known) warnings |= REBX_INPUT_BINARY_WARNING_FIELD_UNKNOWN;
            skip(c, r);
        }
        warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;
    }

    void load_snapshot(Cursor& c) {                                     // input.c:472-529
        Record r;
        if (!next_record(c, r) || r.type != F_SNAPSHOT) { warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; return; }
        while (next_record(c, r)) {
            if (r.type == F_END) return;
            Cursor inner{c.p, c.p + (r.size >= 0 && r.size <= c.end - c.p ? r.size : 0)};
            if (r.type == F_REBX_STRUCTURE) load_structure(inner);
            else if (r.type == F_PARTICLES) { if (!load_list(inner, F_PARTICLE, nullptr)) warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT; }
            else warnings |= REBX_INPUT_BINARY_WARNING_LIST_UNKNOWN;
            skip(c, r);
        }
        warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;
    }
};

std::string header_text() {
    std::string h = "REBOUNDx Binary File. Version: ";
    h += rebx_version_str;
    return h;
}

}  // namespace

extern "C" void rebx_output_binary(struct rebx_extras* rebx, char* filename) {                 // src/output.c:267-288
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return; }
    Writer w;
    // 64-byte header: text, NUL, as much of the git hash as fits, NUL
    const std::string text = header_text();
    w.buf.assign(HEADER, '\0');
    const size_t tl = text.size() < 62 ? text.size() : 62;
    memcpy(&w.buf[0], text.data(), tl);
    const size_t room = 62 - tl, hl = strlen(rebx_githash_str);
    memcpy(&w.buf[tl + 1], rebx_githash_str, hl < room ? hl : room);

    const size_t snapshot = w.open(F_SNAPSHOT);
    const size_t structure = w.open(F_REBX_STRUCTURE);
    {
        const size_t list = w.open(F_REGISTERED_PARAMETERS);
        for_each_oldest_first(rebx->registered_params, [&](void* obj) {
            const struct rebx_param* p = (const struct rebx_param*)obj;
            const size_t rec = w.open(F_REGISTERED_PARAM);
            w.record(F_PARAM_TYPE, &p->type, sizeof(p->type));
            w.text(F_NAME, p->name);
            w.close(rec);
        });
        w.close(list);
    }
    {
        const size_t list = w.open(F_ALLOCATED_FORCES);
        for_each_oldest_first(rebx->allocated_forces, [&](void* obj) { const struct rebx_force* f = (const struct rebx_force*)obj; write_named_with_params(w, rebx, F_FORCE, f->name, f->ap); });
        w.close(list);
    }
    {
        const size_t list = w.open(F_ALLOCATED_OPERATORS);
        for_each_oldest_first(rebx->allocated_operators, [&](void* obj) { const struct rebx_operator* o = (const struct rebx_operator*)obj; write_named_with_params(w, rebx, F_OPERATOR, o->name, o->ap); });
        w.close(list);
    }
    {
        const size_t list = w.open(F_ADDITIONAL_FORCES);
        for_each_oldest_first(rebx->additional_forces, [&](void* obj) {
            const size_t rec = w.open(F_ADDITIONAL_FORCE);
            w.text(F_NAME, ((const struct rebx_force*)obj)->name);
            w.close(rec);
        });
        w.close(list);
    }
    write_steps(w, F_PRE_TIMESTEP_MODIFICATIONS, rebx->pre_timestep_modifications);
    write_steps(w, F_POST_TIMESTEP_MODIFICATIONS, rebx->post_timestep_modifications);
    w.close(structure);
    {
        const size_t list = w.open(F_PARTICLES);
        struct reb_simulation* sim = rebx->sim;
        const ColumnEntries columns(rebx);
        for (int i = 0; i < sim->N; i++) {
            const size_t rec = w.open(F_PARTICLE);
            const int32_t index = i;
            w.record(F_PARTICLE_INDEX, &index, sizeof(index));
            write_param_list(w, rebx, (struct rebx_node*)sim->particles[i].ap, &columns, i);
            w.close(rec);
        }
        w.close(list);
    }
    w.close(snapshot);

    FILE* of = fopen(filename, "wb");
    if (of == NULL) { rebx_error(rebx, "REBOUNDx error: Can not open file passed to rebx_output_binary."); return; }
    const bool ok = fwrite(w.buf.data(), 1, w.buf.size(), of) == w.buf.size();
    if (fclose(of) != 0 || !ok) rebx_error(rebx, "REBOUNDx error: Could not write the file passed to rebx_output_binary.");
}

extern "C" void rebx_init_extras_from_binary(struct rebx_extras* rebx, const char* const filename, enum rebx_input_binary_messages* warnings) {   // src/input.c:627-643
    if (rebx->sim == NULL) { rebx_error(rebx, ""); return; }
    FILE* inf = fopen(filename, "rb");
    if (!inf) { *warnings = (enum rebx_input_binary_messages)(*warnings | REBX_INPUT_BINARY_ERROR_NOFILE); return; }
    std::vector<unsigned char> data;
    unsigned char chunk[65536];
    size_t got;
    while ((got = fread(chunk, 1, sizeof(chunk), inf)) > 0) data.insert(data.end(), chunk, chunk + got);
    fclose(inf);
    Loader ld{rebx};
    // header: the version text must match; the git hash after the first NUL is ignored (input.c:616-625)
    const std::string text = header_text();
    if (data.size() < HEADER || strnlen((const char*)data.data(), HEADER) != text.size() || memcmp(data.data(), text.data(), text.size()) != 0)
        ld.warnings |= REBX_INPUT_BINARY_WARNING_VERSION;
    if (data.size() < HEADER) ld.warnings |= REBX_INPUT_BINARY_ERROR_CORRUPT;
    else {
        Cursor c{data.data() + HEADER, data.data() + data.size()};
        ld.load_snapshot(c);
    }
    *warnings = (enum rebx_input_binary_messages)(*warnings | ld.warnings);
}

extern "C" struct rebx_extras* rebx_create_extras_from_binary(struct reb_simulation* sim, const char* const filename) {    // src/input.c:645-702
    if (sim == NULL) {
        fprintf(stderr, "REBOUNDx Error: Simulation pointer passed to rebx_create_extras_from_binary was NULL.\n");
        return NULL;
    }
    enum rebx_input_binary_messages w = REBX_INPUT_BINARY_WARNING_NONE;
    struct rebx_extras* rebx = (struct rebx_extras*)malloc(sizeof(*rebx));     // no default registrations: the file carries them
    if (!rebx) return NULL;
    rebx_initialize(sim, rebx);
    rebx_init_extras_from_binary(rebx, filename, &w);
    static const struct { int bit; bool error; const char* text; } table[] = {
        {REBX_INPUT_BINARY_ERROR_NOFILE, true, "REBOUNDx: Cannot open binary file. Check filename."},
        {REBX_INPUT_BINARY_ERROR_CORRUPT, true, "REBOUNDx: Binary file is unreadable. Please open an issue on Github mentioning the version of REBOUND and REBOUNDx you are using and include the binary file."},
        {REBX_INPUT_BINARY_ERROR_NO_MEMORY, true, "REBOUNDx: Ran out of system memory."},
        {REBX_INPUT_BINARY_ERROR_REBX_NOT_LOADED, true, "REBOUNDx: REBOUNDx structure couldn't be loaded."},
        {REBX_INPUT_BINARY_ERROR_REGISTERED_PARAM_NOT_LOADED, true, "REBOUNDx: At least one registered parameter was not loaded. This typically indicates the binary is corrupt or was saved with an incompatible version to the current one being used."},
        {REBX_INPUT_BINARY_WARNING_PARAM_NOT_LOADED, false, "REBOUNDx: At least one force or operator parameter was not loaded from the binary file. This typically indicates the binary is corrupt or was saved with an incompatible version to the current one being used."},
        {REBX_INPUT_BINARY_WARNING_PARTICLE_PARAMS_NOT_LOADED, false, "REBOUNDx: At least one particle's parameters were not loaded from the binary file."},
        {REBX_INPUT_BINARY_WARNING_FORCE_NOT_LOADED, false, "REBOUNDx: At least one force was not loaded from the binary file. If binary was created with a newer version of REBOUNDx, a particular force may not be implemented in your current version of REBOUNDx."},
        {REBX_INPUT_BINARY_WARNING_OPERATOR_NOT_LOADED, false, "REBOUNDx: At least one operator was not loaded from the binary file. If binary was created with a newer version of REBOUNDx, a particular force may not be implemented in your current version of REBOUNDx."},
        {REBX_INPUT_BINARY_WARNING_STEP_NOT_LOADED, false, "REBOUNDx: At least one operator step was not loaded from the binary file."},
        {REBX_INPUT_BINARY_WARNING_ADDITIONAL_FORCE_NOT_LOADED, false, "REBOUNDx: At least one force was not added to the simulation. If binary was created with a newer version of REBOUNDx, a particular force may not be implemented in your current version of REBOUNDx."},
        {REBX_INPUT_BINARY_WARNING_FIELD_UNKNOWN, false, "REBOUNDx: Unknown field found in binary file. Any unknown fields not loaded.  This can happen if the binary was created with a later version of REBOUNDx than the one used to read it."},
        {REBX_INPUT_BINARY_WARNING_LIST_UNKNOWN, false, "REBOUNDx: Unknown list in the REBOUNDx structure wasn't loaded. This can happen if the binary was created with a later version of REBOUNDx than the one used to read it."},
        {REBX_INPUT_BINARY_WARNING_PARAM_VALUE_NULL, false, "REBOUNDx: The value of at least one parameter was not loaded. This can happen if a custom structure was added by the user as a parameter. See Parameters.ipynb jupyter notebook example."},
        {REBX_INPUT_BINARY_WARNING_VERSION, false, "REBOUNDx: Binary file was saved with a different version of REBOUNDx. Binary format might have changed. Check that effects and parameters are loaded as expected."},
        {REBX_INPUT_BINARY_WARNING_FORCE_PARAM_NOT_LOADED, false, "REBOUNDx: A force parameter failed to load from the list of REBOUNDx implemented forces. Custom forces can't be saved to a REBOUNDx binary, and function points must be reset when a simulation is reloaded."},
    };
    for (const auto& t : table)
        if (w & t.bit) { if (t.error) reb_error(sim, t.text); else reb_warning(sim, t.text); }
    return rebx;
}
