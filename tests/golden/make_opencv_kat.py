#!/usr/bin/env python3
"""Known-answer vectors for the stages OpenCV computes in the reference
(/root/reference/app/passive_checker.py:100, :107, :111, :116 and cv2.Laplacian at /root/reference/app/main.py:120):
gray conversion, 3x3 high-pass, 5x5 Gaussian, Canny, Laplacian.

Every expected value below was worked out BY HAND from the arithmetic OpenCV documents for its 8-bit paths; nothing
here calls the oracle or the library.  The derivations are kept next to the numbers.  OpenCV itself is not importable
in this image, so these vectors do not pin the restatement to the library; they pin it (C oracle, numpy mirror and HIP
path alike) to an independent reading of the rules at the places where a misreading would show: exact .5 roundings,
the mirrored border at widths 1..3, Canny ties on plateaus, strictness of each comparison, the two direction
boundaries, threshold equalities and swapped thresholds.

    python tests/golden/make_opencv_kat.py     ->  tests/golden/opencv_kat.json
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def plane(h, w, f):
    return [[f(y, x) for x in range(w)] for y in range(h)]


def edges_at(h, w, pts):
    e = [[0] * w for _ in range(h)]
    for (y, x) in pts:
        e[y][x] = 255
    return e


def transpose(m):
    return [list(r) for r in zip(*m)]


cases = []

# ---------------------------------------------------------------------------------------------- gray
# gray = (B*wB + G*wG + R*wR + half) >> shift, round half up.  15-bit weights 3735 / 19235 / 9798 (sum 32768),
# 14-bit weights 1868 / 9617 / 4899 (sum 16384).
cases.append(dict(
    kind="gray", shift=15,
    bgr=[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [4, 12, 0], [186, 0, 1], [0, 47, 90]],
    expect=[255, 0, 29, 150, 76, 8, 21, 55],
    derivation=[
        "(255,255,255): 32768*255 + 16384 >> 15 = 255",
        "(255,0,0): 3735*255 = 952425; +16384 = 968809; 29*32768 = 950272 <= 968809 < 983040 -> 29",
        "(0,255,0): 19235*255 = 4904925; +16384 = 4921309; 150*32768 = 4915200 <= .. < 4947968 -> 150",
        "(0,0,255): 9798*255 = 2498490; +16384 = 2514874; 76*32768 = 2490368 <= .. < 2523136 -> 76",
        "(4,12,0): 14940 + 230820 = 245760 = 7.5 * 32768 exactly: half rounds UP -> 8",
        "(186,0,1): 694710 + 9798 = 704508 = 21*32768 + 16380, the largest remainder below a half that 8-bit inputs "
        "reach (16381..16383 have no solution): 704508 + 16384 = 720892 < 22*32768 = 720896 -> 21",
        "(0,47,90): 904045 + 881820 = 1785865; +16384 = 1802249 >= 55*32768 = 1802240 -> 55 (the 14-bit weights give 54)",
    ]))
cases.append(dict(
    kind="gray", shift=14,
    bgr=[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 144, 140], [1, 35, 160], [0, 47, 90]],
    expect=[255, 29, 150, 76, 127, 68, 54],
    derivation=[
        "(255,0,0): 1868*255 = 476340; +8192 = 484532; 29*16384 = 475136 <= .. < 491520 -> 29",
        "(0,255,0): 9617*255 = 2452335; +8192 = 2460527; 150*16384 = 2457600 <= .. < 2473984 -> 150",
        "(0,0,255): 4899*255 = 1249245; +8192 = 1257437; 76*16384 = 1245184 <= .. < 1261568 -> 76",
        "(1,144,140): 1868 + 1384848 + 685860 = 2072576 = 126.5 * 16384 exactly: half rounds UP -> 127",
        "(1,35,160): 1868 + 336595 + 783840 = 1122303; +8192 = 1130495 = 69*16384 - 1 -> 68",
        "(0,47,90): 451999 + 440910 = 892909; +8192 = 901101 < 55*16384 = 901120 -> 54 (the 15-bit weights give 55)",
    ]))

# ---------------------------------------------------------------------------------------------- Gaussian 5x5
# blur = (S + 128) >> 8, S = sum_ij w_i w_j p, w = [1 4 6 4 1], BORDER_REFLECT_101.
# One bright pixel p at the centre of a 9x9 field of zeros (its 5x5 support stays clear of the border): the output at
# offset (dy,dx) is (w|dy| * w|dx| * p + 128) >> 8 with w0 = 6, w1 = 4, w2 = 1.


def spot(p, vals):
    img = plane(9, 9, lambda y, x: p if (y, x) == (4, 4) else 0)
    out = [[0] * 9 for _ in range(9)]
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            out[4 + dy][4 + dx] = vals[(abs(dy), abs(dx))] if abs(dy) <= abs(dx) else vals[(abs(dx), abs(dy))]
    return img, out


img, out = spot(128, {(0, 0): 18, (0, 1): 12, (0, 2): 3, (1, 1): 8, (1, 2): 2, (2, 2): 1})
cases.append(dict(
    kind="gauss5", gray=img, expect=out, absdiff_sum=222, hp_sum=255, hp_sumsq=65025,
    derivation=[
        "p = 128: 36p = 4608 -> 4736>>8 = 18; 24p = 3072 -> 3200>>8 = 12; 6p = 768 -> 896>>8 = 3; 16p = 2048 -> 2176>>8 = 8; "
        "4p = 512 -> 640>>8 = 2; 1p = 128 -> S + 128 = 256 exactly -> 1 (S = 128 mod 256 rounds up)",
        "sum |gray - blur| = (128 - 18) + 4*12 + 4*3 + 4*8 + 8*2 + 4*1 = 110 + 112 = 222",
        "high-pass: centre 9*128 - 128 = 1024 -> saturates to 255; the eight neighbours 0 - 128 < 0 -> 0; elsewhere 0",
    ]))
img, out = spot(127, {(0, 0): 18, (0, 1): 12, (0, 2): 3, (1, 1): 8, (1, 2): 2, (2, 2): 0})
cases.append(dict(
    kind="gauss5", gray=img, expect=out, absdiff_sum=217, hp_sum=255, hp_sumsq=65025,
    derivation=[
        "p = 127: 4572+128 = 4700>>8 = 18; 3048+128 = 3176>>8 = 12; 762+128 = 890>>8 = 3; 2032+128 = 2160>>8 = 8; "
        "508+128 = 636>>8 = 2; 127+128 = 255>>8 = 0 (S = 127 mod 256 rounds down)",
        "sum |gray - blur| = (127 - 18) + 48 + 12 + 32 + 16 + 0 = 217; high-pass centre 9*127 - 127 = 1016 -> 255",
    ]))
# mirrored border at widths 1, 2, 3 (height 1: every row tap is the row itself, the vertical pass multiplies by 16)
cases.append(dict(
    kind="gauss5", gray=[[201]], expect=[[201]], absdiff_sum=0, hp_sum=0, hp_sumsq=0,
    derivation=["1x1: all 25 taps are the pixel: S = 256 v -> v; high-pass 9v - 9v = 0"]))
cases.append(dict(
    kind="gauss5", gray=[[10, 13]], expect=[[12, 12]], absdiff_sum=3, hp_sum=18, hp_sumsq=324,
    derivation=[
        "1x2 [a b], reflect-101 indices -2..3 -> a b a b a b: both outputs 8a + 8b horizontally, *16 vertically: "
        "S = 128 (a+b) -> (a + b + 1) >> 1 = 24 >> 1 = 12; |10-12| + |13-12| = 3",
        "high-pass: the 3x3 window of pixel 0 holds a three times and b six times: 9a - (3a + 6b) = 6(a - b) = -18 -> 0; "
        "pixel 1: 6(b - a) = 18",
    ]))
cases.append(dict(
    kind="gauss5", gray=[[0, 100, 255]], expect=[[82, 114, 146]], absdiff_sum=82 + 14 + 109, hp_sum=255, hp_sumsq=65025,
    derivation=[
        "1x3 [a b c], indices -2..4 -> c b a b c b a: h0 = 6a + 8b + 2c = 1310, h1 = 4a + 8b + 4c = 1820, "
        "h2 = 2a + 8b + 6c = 2330; blur = (16 h + 128) >> 8 = (h + 8) >> 4: 1318>>4 = 82, 1828>>4 = 114, 2338>>4 = 146",
        "high-pass (three equal rows): 9a - 3(a + 2b) = -600 -> 0; 9b - 3(a + b + c) = -165 -> 0; 9c - 3(2b + c) = 930 -> 255",
    ]))
cases.append(dict(
    kind="gauss5", gray=[[50, 40, 45]], expect=[[44, 44, 43]], absdiff_sum=6 + 4 + 2, hp_sum=90, hp_sumsq=4500,
    derivation=[
        "h0 = 300 + 320 + 90 = 710 -> 718>>4 = 44; h1 = 200 + 320 + 180 = 700 -> 708>>4 = 44; "
        "h2 = 100 + 320 + 270 = 690 -> 698>>4 = 43 (43*16 = 688 <= 698 < 704)",
        "high-pass: 6a - 6b = 60; 6b - 3a - 3c = 240 - 150 - 135 = -45 -> 0; 6c - 6b = 30; sum 90, squares 3600 + 900 = 4500",
    ]))

# ---------------------------------------------------------------------------------------------- Laplacian (aperture 1)
# cv2.Laplacian(gray, CV_64F): kernel [0 1 0; 1 -4 1; 0 1 0], BORDER_REFLECT_101.  On a 1xN image the rows above and
# below are the row itself: lap[x] = p[x-1] + p[x+1] - 2 p[x], with p[-1] = p[1] and p[N] = p[N-2].
cases.append(dict(
    kind="laplacian", gray=[[10, 20, 50, 40]], lap=[[20, 20, -40, 20]], lap_sum=20, lap_sumsq=2800, gray_sumsq=4600,
    derivation=[
        "x=0: 2*20 - 2*10 = 20; x=1: 10 + 50 - 40 = 20; x=2: 20 + 40 - 100 = -40; x=3: 2*50 - 2*40 = 20",
        "sum 20, squares 400 + 400 + 1600 + 400 = 2800 (variance 2800/4 - 5^2 = 675); gray squares 100 + 400 + 2500 + 1600",
    ]))
cases.append(dict(
    kind="laplacian", gray=[[7]], lap=[[0]], lap_sum=0, lap_sumsq=0, gray_sumsq=49,
    derivation=["1x1: every tap is the pixel: 4v - 4v = 0"]))

# ---------------------------------------------------------------------------------------------- Canny
# Sobel with BORDER_REPLICATE, L1 magnitude, magnitude plane padded with zeros, direction from
# |gy| << 15 against |gx| * 13573 (tan 22.5) and that + |gx| << 16 (tan 67.5); horizontal: m > left && m >= right;
# vertical: m > up && m >= down; diagonals strict on both sides; candidate m > low, strong m > high; hysteresis
# 8-connected.


def vstep(step, h=5):
    return plane(h, 6, lambda y, x: 0 if x < 3 else step)


# rows identical: gx(c) = 4 (p[c+1] - p[c-1]) -> magnitudes [0 0 4s 4s 0 0], gy = 0 (horizontal direction).
# column 2: m > left (0) and m >= right (equal) -> kept; column 3: m > left fails (equal) -> dropped.
cases.append(dict(
    kind="canny", gray=vstep(100), low=50, high=150, edges=edges_at(5, 6, [(y, 2) for y in range(5)]),
    derivation=["vertical step of 100: magnitudes 400 400 on columns 2, 3: the plateau keeps its LEFT pixel only "
                "(m > left is strict, m >= right is not); 400 > 150 -> strong -> column 2"]))
cases.append(dict(
    kind="canny", gray=transpose(vstep(100)), low=50, high=150, edges=transpose(edges_at(5, 6, [(y, 2) for y in range(5)])),
    derivation=["horizontal step: the same with rows: the plateau keeps its UPPER row only (m > up strict, m >= down not)"]))
cases.append(dict(
    kind="canny", gray=vstep(10), low=40, high=120, edges=edges_at(5, 6, []),
    derivation=["step 10: m = 40 = low: a candidate needs m > low -> nothing"]))
cases.append(dict(
    kind="canny", gray=vstep(11), low=40, high=120, edges=edges_at(5, 6, []),
    derivation=["step 11: m = 44: candidates, but no strong pixel to grow from -> nothing"]))
cases.append(dict(
    kind="canny", gray=vstep(30), low=40, high=120, edges=edges_at(5, 6, []),
    derivation=["step 30: m = 120 = high: strong needs m > high -> candidates only -> nothing"]))
cases.append(dict(
    kind="canny", gray=vstep(31), low=40, high=120, edges=edges_at(5, 6, [(y, 2) for y in range(5)]),
    derivation=["step 31: m = 124 > high -> column 2"]))
cases.append(dict(
    kind="canny", gray=vstep(40), low=150, high=50, edges=edges_at(5, 6, [(y, 2) for y in range(5)]),
    derivation=["thresholds given the wrong way round are swapped: step 40, m = 160 > 150 -> column 2"]))
cases.append(dict(
    kind="canny", gray=vstep(20), low=150, high=50, edges=edges_at(5, 6, []),
    derivation=["swapped thresholds, step 20: m = 80 is a candidate (> 50) but never strong (> 150) -> nothing"]))

# Planes p = a x + b y on 7x7: gx = 8a inside, 4a on the first and last column (replicated border halves the
# difference); gy = 8b inside, 4b on the first and last row.  The magnitude is constant inside: every interior pixel
# sits on a plateau, and only pixels whose two neighbours along the gradient are border pixels (smaller magnitude) or
# padding (zero) can survive.
P22 = dict(a=2, b=2)
# a = b = 2: inside gx = gy = 16, m = 32; edges 24; corners 16.  |gy|<<15 = 524288 against 16*13573 = 217168 and
# 217168 + 16*65536 = 1265744: diagonal, same sign: neighbours (r-1,c-1), (r+1,c+1), both strict.
#   interior: needs both neighbours off the plateau: only (1,5) [neighbours (0,4) = 24, (2,6) = 24] and (5,1) -> kept, 32
#   top edge (0,x): gx 16, gy 8: 262144 > 217168 -> diagonal: (1,x+1) is 32 for x <= 4, and (1,6) = 24 = m for x = 5 -> none
#   bottom edge (6,x): (5,x-1) is 24 = m for x = 1, else 32 -> none; left / right edge likewise -> none
#   corners: (0,0): (1,1) = 32 -> no; (6,6): (5,5) -> no; (0,6) and (6,0): both neighbours are padding -> kept, 16
# low 10, high 20: strong (1,5), (5,1); weak (0,6), (6,0), each touching a strong one.
cases.append(dict(
    kind="canny", gray=plane(7, 7, lambda y, x: 2 * x + 2 * y), low=10, high=20,
    edges=edges_at(7, 7, [(0, 6), (1, 5), (5, 1), (6, 0)]),
    derivation=["plane 2x + 2y: diagonal direction, same sign; ties 24 > 24 and 32 > 32 are false on both sides "
                "(see the comment block above this case in make_opencv_kat.py)"]))
# opposite signs: p = 2x - 2y + 12: the picture above flipped top to bottom; neighbours (r-1,c+1), (r+1,c-1)
cases.append(dict(
    kind="canny", gray=plane(7, 7, lambda y, x: 2 * x - 2 * y + 12), low=10, high=20,
    edges=edges_at(7, 7, [(0, 0), (1, 1), (5, 5), (6, 6)]),
    derivation=["plane 2x - 2y + 12: gx = 16, gy = -16: the other diagonal; (1,1): neighbours (0,2) = 24 and (2,0) = 24 -> kept"]))
# a = 17, b = 7: 7/17 = 0.41176 < tan 22.5 = 0.41421.  inside gx 136, gy 56, m 192: 56<<15 = 1835008 < 136*13573 = 1845928
# -> HORIZONTAL: m > left && m >= right: column 1 (left neighbour = left edge, 68 + 56 = 124) kept for rows 1..5.
#   top edge (0,x): gx 136, gy 28, m 164: horizontal: (0,1) kept (left = corner 96, right 164 >=); x >= 2: left = 164 -> no
#   bottom edge: (6,1) kept likewise.   corners: gx 68, gy 28: 917504 < 68*13573 = 922964 -> horizontal:
#   (0,0): right (0,1) = 164 > 96 -> no; (0,6): left 164 -> no; same for the lower corners.
#   left edge (y,0): gx 68, gy 56: 1835008 > 922964, < 922964 + 68*65536 = 5379412 -> diagonal: (y+1,1) = 192 or 164 > 124 -> none
#   right edge (y,6): diagonal: (y-1,5) = 164 or 192 -> none
# low 100, high 180: column 1: 164 192 192 192 192 192 164 -> strong in the middle, weak ends attached -> whole column.
cases.append(dict(
    kind="canny", gray=plane(7, 7, lambda y, x: 17 * x + 7 * y), low=100, high=180,
    edges=edges_at(7, 7, [(y, 1) for y in range(7)]),
    derivation=["plane 17x + 7y: |gy|/|gx| = 0.41176 just below tan 22.5: horizontal rule, plateau keeps column 1"]))
# a = 12, b = 5: 5/12 = 0.41667 > tan 22.5.  inside gx 96, gy 40, m 136: 40<<15 = 1310720 > 96*13573 = 1303008 -> DIAGONAL
# (and < 1303008 + 96*65536): as for a = b: interior (1,5) [(0,4) = 96+20 = 116, (2,6) = 48+40 = 88] and (5,1) kept.
#   top edge (0,x): gx 96, gy 20: 655360 < 1303008 -> horizontal: (0,1) kept (left corner 68, right 116 >=); bottom (6,1) too
#   corners: gx 48, gy 20: 655360 > 48*13573 = 651504 -> diagonal (20/48 = 0.41667): (0,0): (1,1) = 136 -> no; (6,6) no;
#   (0,6), (6,0): both neighbours padding -> kept, m = 68.   left / right edges: diagonal, a neighbour of 136 or 116 -> none
# low 60, high 120: strong (1,5), (5,1); weak (0,1), (6,1), (0,6), (6,0).  (0,6)~(1,5), (6,0)~(5,1), (6,1)~(5,1) attach;
# (0,1) touches no survivor -> dropped.
cases.append(dict(
    kind="canny", gray=plane(7, 7, lambda y, x: 12 * x + 5 * y), low=60, high=120,
    edges=edges_at(7, 7, [(1, 5), (0, 6), (5, 1), (6, 0), (6, 1)]),
    derivation=["plane 12x + 5y: |gy|/|gx| = 0.41667 just above tan 22.5: diagonal rule inside, horizontal on the top and "
                "bottom rows (20/96), diagonal again at the corners (20/48); the weak pixel (0,1) has no strong neighbour"]))
# the same two planes transposed exchange gx and gy: 17/7 = 2.42857 > tan 67.5 = 2.41421 -> VERTICAL (m > up && m >= down,
# the mirror image of the horizontal rule), 12/5 = 2.4 < tan 67.5 -> diagonal; the same-sign diagonal rule is symmetric
# under transposition, so the edge maps are the transposes.  Checked at the deciding pixels: inside 7x + 17y:
# 136<<15 = 4456448 > 56*13573 + 56*65536 = 4430104 -> vertical; corner: 68<<15 = 2228224 > 28*79109 = 2215052 -> vertical;
# inside 5x + 12y: 96<<15 = 3145728 < 40*79109 = 3164360 -> diagonal.
cases.append(dict(
    kind="canny", gray=plane(7, 7, lambda y, x: 7 * x + 17 * y), low=100, high=180,
    edges=edges_at(7, 7, [(1, x) for x in range(7)]),
    derivation=["plane 7x + 17y: |gy|/|gx| = 2.42857 just above tan 67.5: vertical rule, plateau keeps row 1"]))
cases.append(dict(
    kind="canny", gray=plane(7, 7, lambda y, x: 5 * x + 12 * y), low=60, high=120,
    edges=edges_at(7, 7, [(5, 1), (6, 0), (1, 5), (0, 6), (1, 6)]),
    derivation=["plane 5x + 12y: |gy|/|gx| = 2.4 just below tan 67.5: diagonal inside, vertical on the left and right "
                "columns; transpose of the 12x + 5y map"]))

with open(os.path.join(HERE, "opencv_kat.json"), "w") as f:
    json.dump({"note": __doc__.strip().split("\n\n")[1].replace("\n", " "), "cases": cases}, f, indent=1)
print(len(cases), "cases")
