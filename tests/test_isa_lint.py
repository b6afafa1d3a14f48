"""The linked library contains no instruction form that is known to compute wrong values beside matrix work on gfx950
(packed-fp32 arithmetic with op_sel:[0,1]; measurements in profiles/r02a_pk_opsel_erratum.txt).  Runs without a GPU."""
from avex_amd import isa_lint


def test_pattern_catches_the_measured_forms_only():
    bad = ["v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_mul_f32 v[10:11], v[18:19], v[28:29] op_sel:[0,1] op_sel_hi:[0,0]",
           "v_pk_fma_f32 v[42:43], v[40:41], v[44:45], v[46:47] op_sel:[0,1,0] op_sel_hi:[1,1,1]",
           "v_pk_add_f32 v[6:7], v[2:3], v[6:7] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"]
    good = ["v_pk_add_f32 v[42:43], v[40:41], v[44:45]",
            "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[1,0]",
            "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[1,1] op_sel_hi:[0,0]",
            "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel_hi:[1,0]",
            "v_pk_fma_f32 v[42:43], v[40:41], v[44:45], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,1,1]",
            "v_pk_mov_b32 v[42:43], v[40:41], v[44:45] op_sel:[0,1]",
            "v_pk_add_f16 v1, v2, v3 op_sel:[0,1]"]
    for i in bad:
        assert isa_lint._BAD.search(i), i
    for i in good:
        assert not isa_lint._BAD.search(i), i


def test_store_hazard_rule():
    """A store of more than 64 bits whose data registers a VALU instruction overwrites less than two wait states later (the form hipcc
    leaves unprotected when the store has a scalar-offset register; measured wrong on gfx950, profiles/r04d_store_hazard.txt)."""
    head = "0000000000001000 <kern>:\n"
    bad = [("\tbuffer_store_dwordx4 v[48:51], v120, s[28:31], s65 offen nt\n\tv_pk_mul_f32 v[48:49], v[56:57], v[56:57]\n", 1),
           ("\tbuffer_store_dwordx4 v[48:51], v120, s[28:31], 0 offen\n\ts_nop 0\n\tv_mov_b32_e32 v50, v56\n", 1),
           ("\tglobal_store_dwordx4 v[4:5], v[0:3], off\n\tv_add_f32_e32 v3, v1, v2\n", 1),
           ("\tglobal_store_dwordx3 v[4:5], v[0:2], off\n\tv_mov_b32_e32 v9, v1\n\tv_mov_b32_e32 v2, v1\n", 1)]
    good = ["\tbuffer_store_dwordx4 v[48:51], v120, s[28:31], 0 offen\n\tv_mov_b32_e32 v172, v174\n\ts_nop 0\n\tv_pk_mul_f32 v[48:49], v[56:57], v[56:57]\n",
            "\tbuffer_store_dwordx4 v[48:51], v120, s[28:31], 0 offen\n\ts_nop 1\n\tv_pk_mul_f32 v[48:49], v[56:57], v[56:57]\n",
            "\tbuffer_store_dwordx2 v[48:49], v120, s[28:31], s65 offen\n\tv_pk_mul_f32 v[48:49], v[56:57], v[56:57]\n",            # 64 bits: no hazard
            "\tglobal_store_dwordx4 v[4:5], v[0:3], off\n\tv_mov_b32_e32 v4, v1\n",                                              # the ADDRESS registers are free at once
            "\tglobal_store_dwordx4 v[4:5], v[0:3], off\n\ts_endpgm\n\tv_mov_b32_e32 v2, v1\n",                                 # not a fall-through
            "\tbuffer_store_dwordx4 v[48:51], v120, s[28:31], 0 offen\n\tv_cmp_lt_f32_e32 vcc, v48, v49\n\tv_readfirstlane_b32 s2, v48\n\tds_read_b128 v[48:51], v9\n"]
    for text, n in bad:
        assert len(isa_lint.find_store_hazards(head + text)) == n, text
    for text in good:
        assert isa_lint.find_store_hazards(head + text) == [], text


def test_library_is_clean(built_lib):
    from avex_amd import _capi
    objs = isa_lint.device_code_objects(_capi.LIB_PATH)
    assert len(objs) >= 8                                   # one bundle per HIP translation unit
    text = isa_lint.disassemble(objs[0])
    assert "s_endpgm" in text                               # the disassembler really ran on device code
    assert isa_lint.find_bad_instructions(_capi.LIB_PATH) == []
