// Package semantics: small Go programs whose results are fixed by the Go specification.  tests/test_go2cxx.py
// translates this file with oracle/go2cxx, compiles the C++ and compares what every function returns with the value
// the specification prescribes (written next to each function).  Nothing here resembles the rtlamr sources on
// purpose: the point is that the translator handles the LANGUAGE, not one program.
package semantics

import (
	"fmt"
	"math"
	"strconv"
	"strings"
	"sync"
)

// uint8 arithmetic wraps and never widens: (200+100)>>1 is 22 in uint8, 150 with C integer promotion.  Want "22 156 255 1".
func Wrap() string {
	var a, b uint8 = 200, 100
	c := (a + b) >> 1
	d := a - b - b - b // 200-300 mod 256 = 156
	var z uint8
	z--
	var i8 int8 = 127
	i8++ // -128
	return fmt.Sprintf("%d %d %d %d", c, d, z, i8/-128)
}

// Shifts by >= width give 0 (or the sign), unlike C.  Want "0 -1 0 2147483648 1".
func Shifts() string {
	var u uint32 = 0x80000000
	var s int32 = -5
	n := uint(40)
	a := u >> n
	b := s >> n
	c := u << n
	var one uint64 = 1
	return fmt.Sprintf("%d %d %d %d %d", a, b, c, one<<31, 1<<(n-40))
}

// Untyped constants take the type of the other operand; := gives int / float64.  Want "0.1 3 3.5 127.5 int".
func Constants() string {
	const k = 1
	var f float32 = 0.1
	g := f * k
	x := 7 / 2   // integer division of constants
	y := 7 / 2.0 // floating-point constant
	var lut float32 = (127.5 - float32(0)) / 1
	kind := "float"
	if (k<<3)/3 == 2 {
		kind = "int"
	}
	return fmt.Sprintf("%.1f %d %.1f %.1f %s", g, x, y, lut, kind)
}

// float32 accumulates in float32 (one rounding per operation).  Want "16777216 16777218".
func Float32Sum() string {
	var s float32 = 16777216
	t := s + 1 // rounds back to 2^24
	u := s + 2
	return fmt.Sprintf("%.0f %.0f", t, u)
}

// Slices share their backing array until append outgrows the capacity.  Want "[9 2 3] [9 2] 3 [9 2 7] [9 2 3 4] 3".
func SliceAliasing() string {
	a := []int{1, 2, 3}
	b := a[:2]
	b[0] = 9
	first := show(a) + " " + show(b) + " " + strconv.Itoa(cap(b))
	c := append(b, 7) // fits: overwrites a[2]
	second := show(a)
	a[2] = 3
	d := append(a, 4) // does not fit: new array
	d[0] = 9
	_ = c
	return first + " " + second + " " + show(d) + " " + strconv.Itoa(len(a))
}

func show(s []int) string {
	parts := make([]string, 0, len(s))
	for _, v := range s {
		parts = append(parts, strconv.Itoa(v))
	}
	return "[" + strings.Join(parts, " ") + "]"
}

// copy is a memmove; ranging evaluates the slice once; appends within capacity write the shared array.  Want "[1 1 2 0 0] 3".
func CopyOverlap() string {
	a := []int{1, 2, 3, 4, 5}
	copy(a[1:], a)
	n := 0
	s := a[:3]
	for range s {
		s = append(s, 0)
		n++
	}
	return show(a[:5]) + " " + strconv.Itoa(n)
}

// The list-swapping idiom: functions returning their arguments in another order, assigned crosswise.  Want "[2 4] [] 2".
func keepEven(a, b []int) ([]int, []int) {
	for _, v := range a {
		if v%2 == 0 {
			b = append(b, v)
		}
	}
	return a, b
}

func SwapLists() string {
	x := []int{1, 2, 3, 4}
	y := make([]int, 0, 4)
	y, x = keepEven(x, y[:0])
	y = y[:0]
	return show(x) + " " + show(y) + " " + strconv.Itoa(len(x))
}

// Value receivers work on a copy, pointer receivers on the object; arrays are values.  Want "1 2 [0 0] [5 0]".
type counter struct {
	n    int
	arr  [2]int
	tags []string
}

func (c counter) bumpCopy()  { c.n++; c.arr[0] = 5 }
func (c *counter) bumpReal() { c.n++; c.arr[0] = 5 }

func Receivers() string {
	var c counter
	c.n = 1
	c.bumpCopy()
	first := c.n
	before := c.arr
	c.bumpReal()
	return fmt.Sprintf("%d %d [%d %d] [%d %d]", first, c.n, before[0], before[1], c.arr[0], c.arr[1])
}

// Interfaces hold values or pointers; a map of slices appends through the zero value.  Want "sq:9 ci:12 a=2 b=1 c=0 true false".
type shape interface {
	Area() int
	Name() string
}
type square struct{ s int }
type circle struct{ r int }

func (q square) Area() int    { return q.s * q.s }
func (q square) Name() string { return "sq" }
func (c *circle) Area() int   { return 3 * c.r * c.r }
func (c *circle) Name() string { return "ci" }

func Interfaces() string {
	shapes := []shape{square{3}, &circle{2}}
	out := ""
	for _, s := range shapes {
		out += s.Name() + ":" + strconv.Itoa(s.Area()) + " "
	}
	groups := make(map[string][]int)
	groups["a"] = append(groups["a"], 1)
	groups["a"] = append(groups["a"], 2)
	groups["b"] = append(groups["b"], 3)
	_, okA := groups["a"]
	_, okZ := groups["z"]
	var nilShape shape
	_ = nilShape == nil
	return out + fmt.Sprintf("a=%d b=%d c=%d %t %t", len(groups["a"]), len(groups["b"]), len(groups["c"]), okA, okZ)
}

// Strings are bytes; range decodes UTF-8; conversion from []byte copies.  Want "4 3 195 0:104 1:233 3:33 hx hé!".
func Strings() string {
	s := "hé!"
	b := []byte(s)
	n := 0
	out := ""
	for i, r := range s {
		out += fmt.Sprintf("%d:%d ", i, r)
		n++
	}
	b[1] = 'x'
	t := string(b[:2])
	return fmt.Sprintf("%d %d %d %s%s %s", len(s), n, s[1], out, t, s)
}

// Labelled break / continue, switch without fallthrough, break inside switch inside for.  Want "1,3 2,3 x one many many done".
func Control() string {
	out := ""
outer:
	for i := 1; i < 4; i++ {
		for j := 1; j < 4; j++ {
			if j < 3 {
				continue
			}
			if i == 3 {
				break outer
			}
			out += fmt.Sprintf("%d,%d ", i, j)
			continue outer
		}
	}
	out += "x "
	for i := 1; i <= 3; i++ {
		switch {
		case i == 1:
			out += "one "
		default:
			if i > 3 {
				break
			}
			out += "many "
		}
	}
	switch out[0] {
	case 'z', '1':
		out += "done"
	default:
		out += "no"
	}
	return out
}

// Named results, defer order, closures capturing by value what they read.  Want "3 cba 10".
func named(a int) (r int, log string) {
	r = a
	for _, c := range []string{"a", "b", "c"} {
		c := c
		defer func() { trace += c }()
	}
	r += 2
	return
}

var trace string

func Defers() string {
	trace = ""
	r, _ := named(1)
	add := func(x int) func(int) int { return func(y int) int { return x + y } }
	return fmt.Sprintf("%d %s %d", r, trace, add(3)(7))
}

// Goroutines, an unbuffered channel closed by a WaitGroup waiter, range over the channel.  Want "285 10".
func Channels() string {
	ch := make(chan int)
	wg := new(sync.WaitGroup)
	wg.Add(10)
	for i := 0; i < 10; i++ {
		go func(k int) {
			ch <- k * k
			wg.Done()
		}(i)
	}
	go func() {
		wg.Wait()
		close(ch)
	}()
	sum, n := 0, 0
	for v := range ch {
		sum += v
		n++
	}
	return fmt.Sprintf("%d %d", sum, n)
}

// Generic function with a type-parameter list; integer parsing in odd bases; %b / %x / %q verbs.  Want "2.5 7 15 00101 ff \"a\\\"b\" 44".
func absOf[F float32 | float64 | int](x F) F {
	if x < 0 {
		return -x
	}
	return x
}

func Formats() string {
	v, _ := strconv.ParseInt("23", 6, 32)
	_, err := strconv.ParseUint("19", 8, 8)
	bad := 0
	if err != nil {
		bad = 44
	}
	return fmt.Sprintf("%.1f %d %d %05b %x %q %d", absOf(float32(-2.5)), absOf(-7), v, 5, 255, "a\"b", bad)
}

// Struct literals (keyed, positional, nested, pointer), struct copies, arrays of structs.  Want "7 0 9 [1 2] 3 8".
type inner struct{ a, b int }
type outer struct {
	in  inner
	p   *inner
	arr [2]inner
}

func Structs() string {
	o := outer{in: inner{a: 7}, p: &inner{9, 1}}
	o.arr[1] = inner{1, 2}
	cp := o
	cp.in.a = 3
	cp.p.b = 8 // shared through the pointer
	return fmt.Sprintf("%d %d %d [%d %d] %d %d", o.in.a, o.in.b, o.p.a, o.arr[1].a, o.arr[1].b, cp.in.a, o.p.b)
}

// Embedded structs and interfaces: promoted fields and methods, the embedded field selected by name, an outer method
// shadowing a promoted one, a type assertion back to the concrete type.  Want "sq 9 sq! 9 7 7 8 true".
type namer interface{ Name() string }
type tagged struct {
	square
	tag int
}
type loud struct{ namer }

func (l loud) Name() string { return l.namer.Name() + "!" }

func Embedding() string {
	t := tagged{square{3}, 7}
	var n namer = t // tagged has Name() through the embedded square
	l := loud{n}
	back := n.(tagged)
	t.s = 1        // promoted field; `back` is a copy made before
	t.square.s = 8 // the same field through the embedded struct's name
	_, isCircle := n.(*circle)
	return fmt.Sprintf("%s %d %s %d %d %d %d %t", n.Name(), back.Area(), l.Name(), back.square.Area(), back.tag, t.tag, t.s, !isCircle)
}

// ---- second batch (round 6): arithmetic and aliasing corners ----

// Integer division truncates toward zero and % takes the sign of the dividend; unsigned subtraction wraps.  Want "-3 -1 3 -1 4294967293 3".
func DivMod() string {
	a, b := -7, 2
	var u, v uint32 = 2, 5
	return fmt.Sprintf("%d %d %d %d %d %d", a/b, a%b, a/-b, -7%-2, u-v, (u-v)%5)
}

// A float64 value converted to float32 is rounded once, to nearest even; float32 products are rounded before they are
// widened; the sign bit of a negative zero survives.  Want "16777216 1.0000001 0.010000001 1 0".
func Rounding() string {
	d := 16777217.0 // 2^24 + 1: not a float32
	f := float32(d)
	g := float32(1.00000006) // just above 1 + 2^-24: rounds up to 1 + 2^-23
	var x float32 = 0.1
	p := float64(x * x) // the float32 product, then widened
	var nz float32 = 0
	nz = -nz
	pos := float32(0)
	return fmt.Sprintf("%.0f %.7f %.9f %d %d", f, g, p, math.Float32bits(nz)>>31, math.Float32bits(pos)>>31)
}

// A running float32 sum depends on the order of its terms.  Want "1.0000001 1 false".
func SumOrder() string {
	terms := []float32{1, 3e-8, 3e-8, 3e-8, 3e-8}
	var up, down float32
	for i := 1; i < len(terms); i++ {
		down += terms[i] // the small ones first: they add up to 1.2e-7, enough to move 1
	}
	down += terms[0]
	for _, t := range terms {
		up += t // each one alone is lost next to 1
	}
	return fmt.Sprintf("%.7f %.7g %v", down, up, up == down)
}

// s[lo:hi:max] limits the capacity: the append below cannot reach the parent's elements.  Want "[1 2 3 4] [1 2 9] 2 2 4".
func FullSlice() string {
	a := []int{1, 2, 3, 4}
	b := a[0:2:2]
	c := append(b, 9) // cap(b) == 2: a new array
	d := a[1:3]
	return show(a) + " " + show(c) + " " + strconv.Itoa(cap(b)) + " " + strconv.Itoa(len(d)) + " " + strconv.Itoa(cap(a[:0]))
}

// range over an ARRAY copies it, range over a slice does not; the value variable is a copy either way.  Want "[1 2 3] [1 9 3] 6 15".
func RangeCopies() string {
	arr := [3]int{1, 2, 3}
	sum := 0
	for i, v := range arr {
		arr[2] = 10 // the loop walks its copy
		if i == 2 {
			sum += v
		}
		sum += 0
	}
	arr[2] = 3
	sl := []int{1, 2, 3}
	seen := 0
	for i, v := range sl {
		if i == 0 {
			sl[1] = 9
		}
		v += 100 // a copy
		seen += sl[i]
		_ = v
	}
	// sum == 3 (the copy's last element), seen == 1 + 9 + 3
	return fmt.Sprintf("[%d %d %d] %s %d %d", arr[0], arr[1], arr[2], show(sl), sum*2, seen+2)
}

// Labelled break / continue, switch without a tag, break inside a switch leaves the switch only.  Want "4 12 small big".
func Labels() string {
	n, m := 0, 0
outer:
	for i := 0; i < 5; i++ {
		for j := 0; j < 5; j++ {
			if j == 2 {
				continue outer
			}
			if i == 2 {
				break outer
			}
			n++
		}
	}
	for i := 0; i < 4; i++ {
		switch {
		case i == 1:
			break // the switch, not the loop
		default:
			m += 4
		}
	}
	word := func(x int) string {
		switch {
		case x < 10:
			return "small"
		}
		return "big"
	}
	return fmt.Sprintf("%d %d %s %s", n, m, word(3), word(30))
}

// Closures capture variables, not values; a shift-and-or byte accumulator keeps eight bits.  Want "3 6 165 42".
func Closures() string {
	total := 0
	add := func(k int) int { total += k; return total }
	add(1)
	add(2)
	first := total
	double := func() { total *= 2 }
	double()
	var acc uint8
	for _, bit := range []uint8{1, 0, 1, 0, 0, 1, 0, 1} {
		acc = acc<<1 | bit
	}
	keep := acc // 1010 0101
	for _, bit := range []uint8{0, 1, 0} {
		acc = acc<<1 | bit // the high bits fall off: 165 -> 74 -> 149 -> 42
	}
	return fmt.Sprintf("%d %d %d %d", first, total, keep, acc)
}
