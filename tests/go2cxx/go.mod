module example.test/semantics

go 1.21
