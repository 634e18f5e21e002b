// Host-only exercise of the plugin API's memory handling (object model, parameter lists,
// operators and steps, binary files, teardown) for a run under AddressSanitizer / LeakSanitizer.
// Nothing here touches a GPU: no force evaluation is requested.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "reboundx_amd.h"

static struct reb_simulation* make_sim(int n) {
    struct reb_simulation* sim = (struct reb_simulation*)calloc(1, sizeof(*sim));
    sim->G = 1.; sim->N = n; sim->dt = 0.1; sim->integrator = REB_INTEGRATOR_WHFAST;
    sim->particles = (struct reb_particle*)calloc((size_t)n, sizeof(struct reb_particle));
    for (int i = 0; i < n; i++) { sim->particles[i].m = i ? 1e-3 : 1.; sim->particles[i].x = i; sim->particles[i].vy = i ? 1. : 0.; sim->particles[i].sim = sim; }
    return sim;
}
// what reb_free_simulation does: particle parameter lists first (through the hook), then the arrays
static void free_sim(struct reb_simulation* sim) {
    for (int i = 0; i < sim->N; i++) rebx_free_particle_ap(&sim->particles[i]);
    rebxamd_clear_messages(sim);
    free(sim->particles);
    free(sim);
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/rebxamd_lifecycle.rebx";
    struct reb_simulation* sim = make_sim(4);
    struct rebx_extras* rebx = rebx_attach(sim);
    rebx_register_param(rebx, "my_count", REBX_TYPE_UINT32);
    struct rebx_force* gr = rebx_load_force(rebx, "gr");
    rebx_set_param_double(rebx, &gr->ap, "c", 1e4);
    rebx_set_param_int(rebx, &gr->ap, "max_iterations", 5);
    rebx_add_force(rebx, gr);
    struct rebx_force* mof = rebx_load_force(rebx, "modify_orbits_forces");
    struct rebx_force* gone = rebx_load_force(rebx, "central_force");
    rebx_add_force(rebx, gone);
    assert(rebx_remove_force(rebx, gone) == 1);
    assert(rebx_load_force(rebx, "no_such_force") == NULL);
    struct rebx_operator* op = rebx_load_operator(rebx, "integrate_force");
    rebx_set_param_pointer(rebx, &op->ap, "force", mof);
    rebx_set_param_int(rebx, &op->ap, "integrator", 1);
    assert(rebx_add_operator(rebx, op) == 1);
    struct rebx_operator* drift = rebx_load_operator(rebx, "drift");
    assert(rebx_add_operator_step(rebx, drift, 0.25, REBX_TIMING_POST) == 1);
    assert(rebx_remove_operator(rebx, drift) == 1);
    assert(rebx_load_operator(rebx, "no_such_operator") == NULL);
    for (int i = 1; i < 4; i++) {
        rebx_set_param_double(rebx, (struct rebx_node**)&sim->particles[i].ap, "tau_a", -1e4 * i);
        rebx_set_param_double(rebx, (struct rebx_node**)&sim->particles[i].ap, "tau_a", -2e4 * i);   // overwrite in place
        rebx_set_param_uint32(rebx, (struct rebx_node**)&sim->particles[i].ap, "my_count", 7u * i);
    }
    rebx_set_param_int(rebx, (struct rebx_node**)&sim->particles[0].ap, "primary", 1);
    {   // a parameter given as a whole column: copied, replaced, refused for names that are not per-particle doubles
        double colv[4] = {__builtin_nan(""), -1., -2., -3.};
        int nn = 0;
        assert(rebxamd_set_param_column(rebx, "tau_e", colv, 4) == 1);
        colv[2] = 9.;
        assert(rebxamd_get_param_column(rebx, "tau_e", &nn)[2] == -2. && nn == 4);
        assert(rebxamd_set_param_column(rebx, "tau_e", colv, 3) == 1 && rebxamd_get_param_column(rebx, "tau_e", &nn)[2] == 9. && nn == 3);
        assert(rebxamd_set_param_column(rebx, "J2", colv, 4) == 0);
        assert(rebxamd_set_param_column(rebx, "tau_inc", colv, 4) == 1 && rebxamd_set_param_column(rebx, "tau_inc", NULL, 0) == 1);
        assert(rebxamd_get_param_column(rebx, "tau_inc", &nn) == NULL && nn == 0);
    }
    assert(*(double*)rebx_get_param(rebx, (struct rebx_node*)sim->particles[2].ap, "tau_a") == -4e4);
    sim->post_timestep_modifications(sim);          // runs the kick-free operator list: integrate_force needs a GPU and reports
    rebx_output_binary(rebx, (char*)path);

    struct reb_simulation* sim2 = make_sim(4);
    struct rebx_extras* rebx2 = rebx_create_extras_from_binary(sim2, path);
    assert(rebx2 && rebx_get_force(rebx2, "gr") && *(double*)rebx_get_param(rebx2, rebx_get_force(rebx2, "gr")->ap, "c") == 1e4);
    assert(rebx_get_operator(rebx2, "integrate_force") != NULL);
    assert(rebx_get_param(rebx2, rebx_get_operator(rebx2, "integrate_force")->ap, "force") == rebx_get_force(rebx2, "modify_orbits_forces"));
    assert(*(uint32_t*)rebx_get_param(rebx2, (struct rebx_node*)sim2->particles[3].ap, "my_count") == 21u);

    // truncated copies of the file must be rejected without touching memory they do not own
    FILE* f = fopen(path, "rb");
    static unsigned char buf[1 << 16];
    const size_t len = fread(buf, 1, sizeof(buf), f);
    fclose(f);
    for (size_t cut = 0; cut < len; cut += 7) {
        char tmp[256];
        snprintf(tmp, sizeof(tmp), "%s.cut", path);
        FILE* g = fopen(tmp, "wb");
        fwrite(buf, 1, cut, g);
        fclose(g);
        struct reb_simulation* s3 = make_sim(4);
        struct rebx_extras* r3 = rebx_create_extras_from_binary(s3, tmp);
        rebx_free(r3);
        free_sim(s3);
        remove(tmp);
    }
    // files that lie about a value: every PARAM_TYPE record (type 6, 4 bytes) is overwritten with every type
    // code in turn, so that 4-byte values claim to be doubles, doubles claim to be force names with no NUL
    // inside, and values claim types nobody frees.  Whatever the loader keeps is then read the way the force
    // evaluation reads it; the sanitizers see any read behind a value and any value left behind.
    for (size_t o = 64; o + 20 <= len; o++) {
        int32_t type; int64_t size;
        memcpy(&type, buf + o, 4); memcpy(&size, buf + o + 8, 8);
        if (type != 6 || size != 4 || memcmp(buf + o + 4, "\0\0\0\0", 4) != 0) continue;
        int32_t original;
        memcpy(&original, buf + o + 16, 4);
        for (int32_t t = 0; t <= 8; t++) {
            if (t == original) continue;
            char tmp[256];
            snprintf(tmp, sizeof(tmp), "%s.lie", path);
            memcpy(buf + o + 16, &t, 4);
            FILE* g = fopen(tmp, "wb");
            fwrite(buf, 1, len, g);
            fclose(g);
            struct reb_simulation* s3 = make_sim(4);
            struct rebx_extras* r3 = rebx_create_extras_from_binary(s3, tmp);
            double sink = 0.;
            auto read_list = [&](struct rebx_node* ap) {
                for (; ap; ap = ap->next) {
                    const struct rebx_param* q = (const struct rebx_param*)ap->object;
                    if (!q->value) continue;
                    if (q->type == REBX_TYPE_DOUBLE) sink += *(const double*)q->value;
                    else if (q->type == REBX_TYPE_INT) sink += *(const int*)q->value;
                    else if (q->type == REBX_TYPE_UINT32) sink += *(const uint32_t*)q->value;
                    else if (q->type == REBX_TYPE_FORCE) sink += (double)strlen(((const struct rebx_force*)q->value)->name);
                }
            };
            for (int i = 0; i < s3->N; i++) read_list((struct rebx_node*)s3->particles[i].ap);
            for (struct rebx_node* n = r3->allocated_forces; n; n = n->next) read_list(((struct rebx_force*)n->object)->ap);
            for (struct rebx_node* n = r3->allocated_operators; n; n = n->next) read_list(((struct rebx_operator*)n->object)->ap);
            if (sink == 12345.678) printf("unlikely\n");
            rebx_free(r3);
            free_sim(s3);
            remove(tmp);
        }
        memcpy(buf + o + 16, &original, 4);
    }
    rebx_free(rebx2);
    free_sim(sim2);
    rebx_free(rebx);
    free_sim(sim);
    remove(path);
    printf("host lifecycle ok\n");
    return 0;
}
